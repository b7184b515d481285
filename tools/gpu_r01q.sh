#!/bin/bash
# Round-1 GPU session Q: rotating-slot blocked kernel (MIFLOW_TB_ROT=1): parity + sweep.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01q
mkdir -p $O; rm -f $O/sweep.jsonl
export MIFLOW_TB_ROT=1
(timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_golden.py -m gpu -q 2>&1 | tail -8) > $O/pytest_tvl1.log
(timeout 300 python tools/sweep_tb.py --no-v1 --tag rot-defaults 2>/dev/null | tail -1) >> $O/sweep.jsonl
for v in "1,4,1:10" "1,4,3:10" "1,5,1:8" "1,6,1:6" "2,2,2:8,10" "2,3,2:4,5,6"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "rot variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep.jsonl
done
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench.json
cat $O/pytest_tvl1.log
python - <<'PY'
import json
for l in open('gpurun_out/r01q/sweep.jsonl'):
    d=json.loads(l); print(d['tag'], {k:round(v['Gpxiter_per_s'],1) for k,v in d.items() if k.startswith('T')})
d=json.loads(open('gpurun_out/r01q/bench.json').read()); print('bench', d['value'], d['roofline']['avg_launch_us'])
PY
