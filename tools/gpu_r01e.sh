#!/bin/bash
# Round-1 GPU session E: TV-L1 gamma tests + interior-specialised tb kernel sweep, full suite, bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01e
mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest_all.log
(timeout 400 python bench.py 2> $O/bench.err | tail -1) > $O/bench.json
for v in "2,1:4,5,6,8" "2,3:5" "1,5:6" "1,4:8,10" "2,4:3,4" "1,6:5,6" "1,8:3,4"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep_variants.jsonl
done
(timeout 200 python bench.py --workload surf --batch 2 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1) > $O/surf_bench.json
find $O -type f -size +4M -delete
ls -la $O
