#!/bin/bash
# Round-1 GPU session K: full GPU suite on the final code, default bench line, secondary workloads, kernel trace + PMC.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01k
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/pytest_gpu.log
(timeout 400 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
(timeout 200 python bench.py --workload stereobm --no-cpu 2>/dev/null | tail -1) > $O/stereobm_bench.json
(timeout 200 python bench.py --workload farneback --no-cpu 2>/dev/null | tail -1) > $O/farneback_bench.json
(timeout 200 python bench.py --workload surf --batch 2 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1) > $O/surf_bench.json
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace -- python $R/bench.py --no-variants --no-cpu > $R/$O/trace_bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/$O/pmc_fetch -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $R/$O/pmc_write -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.md 2>$O/pmc_summary.err
find $O -type f -size +4M -delete
ls -laR $O | head -60
