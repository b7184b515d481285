#!/bin/bash
# Round-1 GPU session D: SURF parity + bench, TV-L1 exact-block sweep + bench, full gpu suite.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01d
mkdir -p $O
(timeout 900 python -m pytest tests/test_surf.py -m gpu -q 2>&1 | tail -60) > $O/pytest_surf.log
(timeout 600 python -m pytest tests -m gpu -x -q --deselect tests/test_surf.py 2>&1 | tail -15) > $O/pytest_rest.log
(timeout 300 python bench.py --workload surf --batch 2 --steps 2 --warmup 1 2> $O/surf.err | tail -1) > $O/surf_bench.json
(timeout 400 python bench.py 2> $O/bench.err | tail -1) > $O/bench.json
for v in "2,1:4,5,8" "2,3:5" "1,5:6" "1,4:8,10" "2,4:3,4"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep_variants.jsonl
done
(timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace_surf -- python bench.py --workload surf --batch 1 --steps 2 --warmup 1 --no-cpu > $O/ktrace_surf.log 2>&1)
find $O/ktrace_surf -name "*kernel_stats.csv" -exec cp {} $O/surf_kernel_stats.csv \;
find $O -type f -size +4M -delete
ls -la $O
