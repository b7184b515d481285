#!/bin/bash
# Round-1 GPU session H: 4-px warp kernel A/B, batch sweep, SURF descriptor changes, full suite.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01h
mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest_all.log
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_warp4.json
(MIFLOW_WARP1=1 timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_warp1.json
for b in 32 48 64; do
  (timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-variants --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('batch=$b', j['value'])") >> $O/batch_sweep.log 2>&1
done
(timeout 200 python bench.py --workload surf --batch 2 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1) > $O/surf_bench.json
BCMD="python bench.py --steps 2 --warmup 1 --no-variants --no-cpu"
(timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace -- $BCMD > $O/ktrace.log 2>&1)
find $O/ktrace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
(timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace_surf -- python bench.py --workload surf --batch 1 --steps 2 --warmup 1 --no-cpu > $O/ktrace_surf.log 2>&1)
find $O/ktrace_surf -name "*kernel_stats.csv" -exec cp {} $O/surf_kernel_stats.csv \;
find $O -type f -size +4M -delete
ls -la $O
