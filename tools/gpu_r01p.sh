#!/bin/bash
# Round-1 GPU session P: blocked kernel with interior (straight-line) / edge step specialisation + LDS static prefetch.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01p
mkdir -p $O
export MIFLOW_TB_SWZ=1
(timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_golden.py -m gpu -q 2>&1 | tail -5) > $O/pytest_tvl1.log
(timeout 300 python tools/sweep_tb.py --no-v1 --tag defaults 2>/dev/null | tail -1) >> $O/sweep.jsonl
for v in "1,4,1:8" "1,3,2:8,10" "1,3,4:8" "2,1,1:4,8,10" "2,2,2:5" "1,5,2:5" "1,4,4:5,6" "2,3,2:3,4" "1,6,2:4" "1,5,4:4"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep.jsonl
done
cat $O/pytest_tvl1.log
python - <<'PY'
import json
for l in open('gpurun_out/r01p/sweep.jsonl'):
    d=json.loads(l); print(d['tag'], {k:round(v['Gpxiter_per_s'],1) for k,v in d.items() if k.startswith('T')})
PY
