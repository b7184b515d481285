#!/bin/bash
# Round-1 GPU session G: full suite after SURF/StereoBM optimisations, benches, PMC traffic of the default TV-L1 path.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01g
mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest_all.log
(timeout 200 python bench.py --workload stereobm --batch 8 --steps 3 --warmup 1 2>/dev/null | tail -1) > $O/sbm_bench.json
(timeout 200 python bench.py --workload surf --batch 2 --steps 2 --warmup 1 2>/dev/null | tail -1) > $O/surf_bench.json
(timeout 200 python bench.py --workload farneback --batch 20 --steps 3 --warmup 1 2>/dev/null | tail -1) > $O/fb_bench.json
(timeout 500 python bench.py 2> $O/bench.err | tail -1) > $O/bench.json
for b in 4 8 32; do
  (timeout 200 python bench.py --batch $b --steps 3 --warmup 1 --no-variants --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('batch=$b', j['value'])") >> $O/batch_sweep.log 2>&1
done
BCMD="python bench.py --steps 2 --warmup 1 --no-variants --no-cpu"
(timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace -- $BCMD > $O/ktrace.log 2>&1)
find $O/ktrace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --pmc $c -f csv -d $O/pmc_$c -- python bench.py --steps 1 --warmup 0 --no-variants --no-cpu > $O/pmc_$c.log 2>&1)
  python tools/pmc_summary.py $O/pmc_$c > $O/pmc_$c.md 2>/dev/null
done
(timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace_surf -- python bench.py --workload surf --batch 1 --steps 2 --warmup 1 --no-cpu > $O/ktrace_surf.log 2>&1)
find $O/ktrace_surf -name "*kernel_stats.csv" -exec cp {} $O/surf_kernel_stats.csv \;
(timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace_sbm -- python bench.py --workload stereobm --batch 4 --steps 2 --warmup 1 --no-cpu > $O/ktrace_sbm.log 2>&1)
find $O/ktrace_sbm -name "*kernel_stats.csv" -exec cp {} $O/sbm_kernel_stats.csv \;
find $O -type f -size +4M -delete
ls -la $O
