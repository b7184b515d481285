#!/bin/bash
# Round-2 opening GPU session (prepared at the end of round 1, when the GPU budget was spent): first execution of what was written
# blind -- the CPU-class SURF kernels, sparse PyrLK, the SURF acceptance tests against the CPU class -- then the state of record again.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r02a.sh'
set -u
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
(timeout 120 python -m pytest tests/test_zz_surf_cpu_class.py tests/test_zzz_surf_cpu_class_hip.py tests/test_zzz_sparse_pyrlk_hip.py tests/test_zzz_bfmatch_int_hip.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25) > $O/pytest_new.log
(timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12) > $O/pytest_gpu.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4) > $O/smoke.log
(timeout 300 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
(timeout 60 python tools/bf_timing.py 2>&1 | tail -3) > $O/bf_timing.log
(timeout 120 python bench.py --workload farneback --no-cpu 2>/dev/null | tail -1) > $O/farneback_bench.json
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace -- python $R/bench.py --no-variants --no-cpu > $R/$O/trace_bench.log 2>&1
cd $R
find $O -type f -size +4M -delete
cat $O/pytest_new.log $O/pytest_gpu.log $O/smoke.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02a/bench.json').read())
print('tvl1', d['value'], d['roofline']['avg_launch_us'], {k: (round(v['pairs_per_s'], 1) if 'pairs_per_s' in v else v) for k, v in d.get('variants', {}).items()})
PY
