#!/bin/bash
# Round-1 GPU session O: timing-only A/B -- the blocked kernel WITHOUT its border fix-up branches (libmiflow_nb.so, wrong at
# the image borders) vs the product build: how much the per-stage scalar branches cost.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01o
mkdir -p $O
for lib in libmiflow.so libmiflow_nb.so; do
  (MIFLOW_LIB=$lib MIFLOW_TB_SWZ=1 timeout 300 python tools/sweep_tb.py --blocks 4,5,6,8,10 --no-v1 --tag lib=$lib 2>/dev/null | tail -1) >> $O/sweep.jsonl
  for v in "1,4,1:8" "1,3,2:8,10" "2,1,1:8,10" "2,2,2:5" "1,5,2:5"; do
    var=${v%%:*}; blocks=${v##*:}
    (MIFLOW_LIB=$lib MIFLOW_TB_SWZ=1 MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "lib=$lib variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r01o/sweep.jsonl'):
    d=json.loads(l); print(d['tag'], {k:round(v['Gpxiter_per_s'],1) for k,v in d.items() if k.startswith('T')})
PY
