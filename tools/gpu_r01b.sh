#!/bin/bash
# Round-1 GPU session B: StereoBM parity + timing, TV-L1 variant sweep, full gpu suite, bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01b
mkdir -p $O
(timeout 600 python -m pytest tests/test_stereobm.py -m gpu -x -q 2>&1 | tail -25) > $O/pytest_sbm.log
(timeout 300 python bench.py --workload stereobm --batch 8 --steps 3 --warmup 1 2> $O/sbm.err | tail -1) > $O/sbm_bench.json
for rb in 16 24 32 48 64; do
  (MIFLOW_SBM_ROWS=$rb timeout 120 python bench.py --workload stereobm --batch 8 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('rb=$rb', j['value'], j['pixel_disparities_per_s'])") >> $O/sbm_rows.log 2>&1
done
# TV-L1 temporal-blocking variants: ppl,wps
for v in "2,1:3,4,5,6,8" "2,4:3,4" "2,3:5,6" "1,8:2,3,4,5" "1,6:5,6" "1,5:6,8" "1,4:8,10" "1,3:10"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep_variants.jsonl
done
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest_all.log
(timeout 400 python bench.py 2> $O/bench.err | tail -1) > $O/bench.json
(timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace_sbm -- python bench.py --workload stereobm --batch 4 --steps 2 --warmup 1 --no-cpu > $O/ktrace_sbm.log 2>&1)
find $O/ktrace_sbm -name "*kernel_stats.csv" -exec cp {} $O/sbm_kernel_stats.csv \;
find $O -type f -size +4M -delete
ls -la $O
