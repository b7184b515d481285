#!/bin/bash
# Round-1 GPU session C: Farneback parity, full gpu suite, TV-L1 bench with packed warp + new variants, profiles.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01c
mkdir -p $O
(timeout 900 python -m pytest tests/test_farneback.py -m gpu -q 2>&1 | tail -40) > $O/pytest_fb.log
(timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_farneback.py 2>&1 | tail -15) > $O/pytest_rest.log
(timeout 400 python bench.py 2> $O/bench.err | tail -1) > $O/bench.json
(timeout 200 python bench.py --workload farneback --batch 20 --steps 3 --warmup 1 2>/dev/null | tail -1) > $O/fb_bench.json
(timeout 200 python bench.py --workload farneback --width 1920 --height 1088 --batch 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1) > $O/fb_bench_1080.json
(timeout 200 python bench.py --batch 64 --steps 3 --warmup 1 --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_b64.json
BCMD="python bench.py --steps 2 --warmup 1 --no-variants --no-cpu"
(timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace -- $BCMD > $O/ktrace.log 2>&1)
find $O/ktrace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --pmc $c -f csv -d $O/pmc_$c -- python bench.py --steps 1 --warmup 0 --no-variants --no-cpu > $O/pmc_$c.log 2>&1)
  python tools/pmc_summary.py $O/pmc_$c > $O/pmc_$c.md 2>/dev/null
done
(timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace_fb -- python bench.py --workload farneback --batch 5 --steps 2 --warmup 1 --no-cpu > $O/ktrace_fb.log 2>&1)
find $O/ktrace_fb -name "*kernel_stats.csv" -exec cp {} $O/fb_kernel_stats.csv \;
find $O -type f -size +4M -delete
ls -la $O
