#!/bin/bash
# Round-1 GPU session U (final): full GPU suite, smoke, bench of record, kernel trace, HBM counters, secondary workloads.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01u
mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/pytest_gpu.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4) > $O/smoke.log
(timeout 300 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace -- python $R/bench.py --no-variants --no-cpu > $R/$O/trace_bench.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/$O/pmc_fetch -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -f csv -d $R/$O/pmc_write -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.md 2>$O/pmc_summary.err
(timeout 200 python bench.py --workload stereobm 2>/dev/null | tail -1) > $O/stereobm_bench.json
(timeout 120 python bench.py --workload farneback 2>/dev/null | tail -1) > $O/farneback_bench.json
(timeout 120 python bench.py --workload surf --batch 2 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1) > $O/surf_bench.json
find $O -type f -size +4M -delete
cat $O/pytest_gpu.log $O/smoke.log
python - <<'PY'
import json
O='gpurun_out/r01u/'
d=json.loads(open(O+'bench.json').read()); print('tvl1', d['value'], d['roofline']['avg_launch_us'], {k:round(v['pairs_per_s'],1) for k,v in d['variants'].items()}, d['cpu_baseline']['value'])
for f in ['stereobm','farneback','surf']:
    try:
        d=json.loads(open(O+f+'_bench.json').read()); print(f, {k:v for k,v in d.items() if isinstance(v,(int,float)) and k not in ('n_gpus','steps','warmup')}, d.get('cpu_baseline'))
    except Exception as e: print(f, 'ERR', e)
PY
head -7 $O/pmc_summary.md; head -6 $O/trace/*/*kernel_stats.csv
