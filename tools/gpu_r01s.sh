#!/bin/bash
# Round-1 GPU session S: planner occupancy assumption vs measured throughput for the rotating-slot kernels.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01s
mkdir -p $O; rm -f $O/sweep.jsonl
export MIFLOW_TB_ROT=1
for cfg in "2:1,4,2:10" "3:1,4,2:10" "2:1,4,1:10" "3:1,4,1:10" "2:1,4,2:8" "3:1,4,2:8" "4:1,5,1:8" "3:1,5,1:8" "3:1,5,2:6" "4:1,5,2:6" "4:1,6,2:5" "5:1,6,2:5" "3:1,6,2:5" "4:1,7,2:4" "5:1,7,2:4" "6:1,7,2:4" "4:1,8,2:3" "6:1,8,2:3" "8:1,8,2:2" "5:1,8,2:2"; do
  w=${cfg%%:*}; rest=${cfg#*:}; var=${rest%%:*}; blocks=${rest##*:}
  (MIFLOW_TB_WPS=$w MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --reps 7 --tag "rot wps=$w variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r01s/sweep.jsonl'):
    d=json.loads(l); print(d['tag'], {k:round(v['Gpxiter_per_s'],1) for k,v in d.items() if k.startswith('T')})
PY
