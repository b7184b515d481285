#!/bin/bash
# Round-1 GPU session L: unconditional (clamped) prefetch loads in the temporally blocked kernel -- parity + sweep + bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01l
mkdir -p $O
(timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_golden.py -m gpu -q 2>&1 | tail -15) > $O/pytest_tvl1.log
(timeout 300 python tools/sweep_tb.py --no-v1 --tag defaults 2>/dev/null | tail -1) > $O/sweep.jsonl
for v in "2,3,1:5" "2,3,2:4,5" "2,2,2:5" "1,5,2:5,6" "1,4,4:5,6" "1,4,1:8,10" "1,3,2:8,10" "1,3,4:8" "2,1,1:8,10"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep.jsonl
done
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench.json
cat $O/pytest_tvl1.log; cat $O/sweep.jsonl; cat $O/bench.json
