#!/bin/bash
# Round-1 GPU session F: StereoBM packed winner-take-all, TV-L1 gamma fix + packed-math (SLP) A/B, full suite.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01f
mkdir -p $O
(timeout 600 python -m pytest tests/test_stereobm.py -m gpu -q 2>&1 | tail -25) > $O/pytest_sbm.log
(timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_stereobm.py 2>&1 | tail -15) > $O/pytest_rest.log
(timeout 200 python bench.py --workload stereobm --batch 8 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1) > $O/sbm_bench.json
for rb in 12 16 24 32; do
  (MIFLOW_SBM_ROWS=$rb timeout 120 python bench.py --workload stereobm --batch 8 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('rb=$rb', j['value'], j['pixel_disparities_per_s'])") >> $O/sbm_rows.log 2>&1
done
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_default.json
(MIFLOW_LIB=libmiflow_slp.so timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_slp.json
(MIFLOW_LIB=libmiflow_slp.so MIFLOW_TB_VARIANT=2,1 timeout 200 python tools/sweep_tb.py --blocks 4,5,8 --tag "slp variant=2,1" --no-v1 2>/dev/null | tail -1) >> $O/sweep_slp.jsonl
(MIFLOW_LIB=libmiflow_slp.so MIFLOW_TB_VARIANT=2,3 timeout 200 python tools/sweep_tb.py --blocks 5 --tag "slp variant=2,3" --no-v1 2>/dev/null | tail -1) >> $O/sweep_slp.jsonl
(MIFLOW_LIB=libmiflow_slp.so MIFLOW_TB_VARIANT=1,4 timeout 200 python tools/sweep_tb.py --blocks 8 --tag "slp variant=1,4" --no-v1 2>/dev/null | tail -1) >> $O/sweep_slp.jsonl
(timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace_sbm -- python bench.py --workload stereobm --batch 4 --steps 2 --warmup 1 --no-cpu > $O/ktrace_sbm.log 2>&1)
find $O/ktrace_sbm -name "*kernel_stats.csv" -exec cp {} $O/sbm_kernel_stats.csv \;
find $O -type f -size +4M -delete
ls -la $O
