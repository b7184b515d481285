#!/bin/bash
# Round-1 GPU session J: prefetch-depth sweep of the temporally blocked kernel, TV-L1 tests (median etc.).
set -u
export TMPDIR=/tmp
O=gpurun_out/r01j
mkdir -p $O
(timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -15) > $O/pytest_tvl1.log
for v in "2,3,1:5" "2,3,2:3,4,5" "2,2,2:5" "1,6,2:4" "1,5,2:5,6" "1,4,2:6,8" "1,3,2:8,10" "1,5,4:4" "1,4,4:5,6" "1,3,4:8,10" "1,6,4:3" "2,3,4:3" "2,2,4:4"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep_pf.jsonl
done
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench.json
(timeout 200 python bench.py --workload surf --batch 2 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1) > $O/surf_bench.json
find $O -type f -size +4M -delete
ls -la $O
