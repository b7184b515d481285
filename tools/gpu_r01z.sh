#!/bin/bash
# Round-1 GPU session Z (last seconds of the budget): matcher tests with the four-wave shape as the default.
set -u
O=gpurun_out/r01z
mkdir -p $O
(timeout 22 python -m pytest tests/test_bfmatch.py tests/test_golden.py -m gpu -q -p no:cacheprovider -k "bf or match or single_wave" 2>&1 | tail -8) > $O/pytest_bf.log
cat $O/pytest_bf.log
