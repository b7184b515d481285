#!/bin/bash
# Round-1 GPU session Y: full GPU suite with the extended brute-force matcher, smoke, matcher timings.
set -u
O=gpurun_out/r01y
mkdir -p $O
(timeout 85 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25) > $O/pytest_gpu.log
(timeout 30 python tools/bf_timing.py 2>&1 | tail -5) > $O/bf_timing.log
cat $O/pytest_gpu.log $O/bf_timing.log
