#!/bin/bash
# Round-1 GPU session T: rotating-slot kernels as default -- full GPU suite, bench of record, kernel trace, HBM counters.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01t
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/pytest_gpu.log
(timeout 400 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace -- python $R/bench.py --no-variants --no-cpu > $R/$O/trace_bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/$O/pmc_fetch -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $R/$O/pmc_write -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.md 2>$O/pmc_summary.err
find $O -type f -size +4M -delete
cat $O/pytest_gpu.log; python -c "
import json; d=json.loads(open('$O/bench.json').read()); print(d['value'], d['roofline']['avg_launch_us'], {k:v['pairs_per_s'] for k,v in d['variants'].items()})"
head -8 $O/pmc_summary.md; cat $O/trace/*/*kernel_stats.csv | head -8
