#!/usr/bin/env bash
# A/B of build variants of the library on ONE box, alternating (box-to-box spread is ~5 %): the TV-L1 headline step per variant.
#   usage: bash tools/ab_libs.sh <out_dir> <rounds> <variant> [<variant> ...]      variant "rel" = libmiflow.so, "x" = libmiflow_x.so;
#   a variant may carry bench arguments and environment after a colon, e.g. "rel:--lanes 3" or "rel:MIFLOW_TB_JW=4"
set -u
OUT=$1; R=$2; shift 2
mkdir -p "$OUT"
for r in $(seq 1 "$R"); do
  for v in "$@"; do
    name=${v%%:*}; extra=""; [ "$v" != "$name" ] && extra=${v#*:}
    lib=libmiflow.so; [ "$name" != "rel" ] && lib=libmiflow_$name.so
    envs=""; args=""
    for tok in $extra; do case "$tok" in *=*) envs="$envs $tok";; *) args="$args $tok";; esac; done
    line=$(env MIFLOW_LIB=$lib $envs timeout 600 python bench.py --no-variants --no-cpu --no-secondary --no-power --steps 10 --warmup 3 $args 2>>"$OUT/ab_err.log" | grep '^{' | tail -1)
    val=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.1f pairs/s  %.2f ms/step  iterate %.0f us' % (d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us') or 0))" "$line" 2>/dev/null)
    echo "round $r  $v: $val" | tee -a "$OUT/ab.log"
  done
done
