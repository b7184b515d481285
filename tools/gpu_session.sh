#!/bin/bash
# One GPU session: `gpurun --timeout N -- 'bash tools/gpu_session.sh <name> <steps...>'`; results under gpurun_out/<name>/.
# Steps: tests_new | tests_all | smoke | ab_warp | bench | trace | pmc | secondary | surf_bench
set -u
export TMPDIR=/tmp
NAME=$1; shift
O=gpurun_out/$NAME
mkdir -p $O
R=$PWD
for step in "$@"; do
case $step in
tests_tile)
  (timeout 600 python -m pytest tests/test_tvl1_gpu.py tests/test_baseline_sizes.py -m gpu -q -x -p no:cacheprovider -k "tile or warp or calc_fast or iterate_blocked or tvl1" 2>&1 | tail -15) > $O/pytest_tile.log; cat $O/pytest_tile.log ;;
ab_tile)
  # register-tile kernel on the small levels: threshold (pixels x pairs per lane) and variant
  CFGS="${AB_TILE_CFGS:-0,0 300000,0 1200000,0 2300000,0 2300000,2 9000000,0 9000000,2 40000000,2}"
  for cfg in $CFGS; do
    mx=${cfg%,*}; v=${cfg#*,}
    (MIFLOW_TILE_MAXPX=$mx MIFLOW_TILE_VARIANT=$v timeout 300 python bench.py --no-variants --no-cpu --no-secondary --steps 12 --warmup 3 2>$O/ab_tile_${mx}_$v.err | tail -1) > $O/ab_tile_${mx}_$v.json
    python - <<PY
import json
try:
    d = json.loads(open('$O/ab_tile_${mx}_$v.json').read()); r = d['roofline']
    print('ab_tile maxpx=$mx variant=$v', round(d['value'], 1), 'pairs/s', round(d['ms_per_step'], 2), 'ms/step | iterate us', round(r['avg_launch_us'], 1), 'warp us', round(r['second_kernel']['avg_launch_us'], 1), 'epe', d.get('epe_vs_analytic_flow_px'))
except Exception as e: print('ab_tile $mx $v failed', e); print(open('$O/ab_tile_${mx}_$v.err').read()[-2000:])
PY
  done ;;
timeline)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace -f csv -d $R/$O/tl -- python $R/bench.py --no-variants --no-cpu --no-secondary --steps 3 --warmup 2 > $R/$O/tl_bench.log 2>&1
  cd $R
  python tools/timeline.py $O/tl > $O/timeline.txt 2>&1; tail -5 $O/timeline.txt
  find $O -type f -size +4M -delete ;;
ab_env)
  # generic A/B over environment settings: AB_ENVS="A=1+B=2 A=0 -" (one bench run per word; "-" = defaults), AB_ARGS = extra bench args
  i=0
  for cfg in ${AB_ENVS:--}; do
    i=$((i+1))
    envs=""; [ "$cfg" != "-" ] && envs=$(echo $cfg | tr '+' ' ')
    (env $envs timeout 300 python bench.py --no-variants --no-cpu --no-secondary --steps ${AB_STEPS:-12} --warmup 3 ${AB_ARGS:-} 2>$O/ab_env_$i.err | tail -1) > $O/ab_env_$i.json
    python - <<PY
import json
try:
    d = json.loads(open('$O/ab_env_$i.json').read()); r = d['roofline']
    print('ab_env [$cfg] ${AB_ARGS:-}', round(d['value'], 1), 'pairs/s', round(d['ms_per_step'], 2), 'ms/step | iterate us', round(r['avg_launch_us'], 1), 'warp us', round(r['second_kernel']['avg_launch_us'], 1), 'epe', d.get('epe_vs_analytic_flow_px'))
except Exception as e: print('ab_env [$cfg] failed', e); print(open('$O/ab_env_$i.err').read()[-2000:])
PY
  done ;;
ab_args)
  # one bench run per ';'-separated argument list in AB_ARGLIST (environment from AB_ENV, space separated)
  i=0
  IFS=';' read -ra LISTS <<< "${AB_ARGLIST:-}"
  for a in "${LISTS[@]}"; do
    i=$((i+1))
    (env ${AB_ENV:-} timeout 300 python bench.py --no-variants --no-cpu --no-secondary --warmup 3 $a 2>$O/ab_args_$i.err | tail -1) > $O/ab_args_$i.json
    python - <<PY
import json
try:
    d = json.loads(open('$O/ab_args_$i.json').read()); r = d['roofline']
    print('ab_args [$a] ${AB_ENV:-}', round(d['value'], 1), 'pairs/s', round(d['ms_per_step'], 2), 'ms/step | iterate us', round(r['avg_launch_us'], 1), 'warp us', round(r['second_kernel']['avg_launch_us'], 1), 'epe', d.get('epe_vs_analytic_flow_px'))
except Exception as e: print('ab_args [$a] failed', e); print(open('$O/ab_args_$i.err').read()[-2000:])
PY
  done ;;
tests_new)
  (timeout 900 python -m pytest tests/test_baseline_sizes.py tests/test_ref_pin.py tests/test_tvl1_gpu.py tests/test_superres_flowio.py tests/test_cpp_shim.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30) > $O/pytest_new.log; cat $O/pytest_new.log ;;
tests_all)
  (timeout 1500 python -m pytest tests -m gpu -q -rf -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_gpu.log; cat $O/pytest_gpu.log ;;
smoke)
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6) > $O/smoke.log; cat $O/smoke.log ;;
ab_warp)
  for cfg in "1 2" "2 2" "4 2" "4 1"; do
    set -- $cfg
    (MIFLOW_WARP_NP=$1 timeout 300 python bench.py --no-variants --no-cpu --no-secondary --lanes $2 --steps 12 --warmup 3 2>$O/ab_np$1_l$2.err | tail -1) > $O/ab_np$1_l$2.json
    python - <<PY
import json
try:
    d = json.loads(open('$O/ab_np$1_l$2.json').read()); r = d['roofline']
    print('ab np=$1 lanes=$2', round(d['value'], 1), 'pairs/s', round(d['ms_per_step'], 2), 'ms/step | iterate us', round(r['avg_launch_us'], 1), 'warp us', round(r['second_kernel']['avg_launch_us'], 1))
except Exception as e: print('ab $1 $2 failed', e); print(open('$O/ab_np$1_l$2.err').read()[-2000:])
PY
  done ;;
bench)
  # stdout = the compact line the driver parses (bench_line.json); the long form is bench_full.json (copied as bench.json)
  (timeout 900 python bench.py ${BENCH_ARGS:-} 2>$O/bench.err) > $O/bench_stdout.txt; tail -1 $O/bench_stdout.txt > $O/bench_line.json
  echo "stdout lines: $(wc -l < $O/bench_stdout.txt), last line bytes: $(tail -1 $O/bench_stdout.txt | wc -c)"
  cp bench_full.json $O/bench.json 2>/dev/null; cat $O/bench_line.json; grep -v "full record" $O/bench.err | tail -5 ;;
trace)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace -- python $R/bench.py --no-variants --no-cpu --no-secondary --steps 5 --warmup 2 > $R/$O/trace_bench.log 2>&1
  cd $R
  find $O/trace -name "*kernel_stats.csv" | head -3
  for f in $(find $O/trace -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats.csv; head -12 $f; done
  find $O -type f -size +4M -delete ;;
pmc)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$O/pmc_$c -- python $R/bench.py --no-variants --no-cpu --no-secondary --steps 2 --warmup 1 > $R/$O/pmc_$c.log 2>&1
  done
  cd $R
  python tools/pmc_summary.py $O > $O/pmc_summary.md 2>&1; head -30 $O/pmc_summary.md
  python tools/pmc_to_traffic.py $O "profiles/$NAME/pmc_summary.md" ${PMC_BATCH:-64} > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -40 $O/pmc_traffic.log
  find $O -type f -size +4M -delete ;;
trace_defaults)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace_defaults -- python $R/bench.py --defaults --no-variants --no-cpu --no-secondary --steps 2 --warmup 1 > $R/$O/trace_defaults_bench.log 2>&1
  cd $R
  tail -1 $O/trace_defaults_bench.log | cut -c1-600
  for f in $(find $O/trace_defaults -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats_defaults.csv; head -12 $f; done
  find $O -type f -size +4M -delete ;;
tests_batch)
  (timeout 600 python -m pytest tests/test_farneback.py tests/test_stereobm.py tests/test_cpp_shim.py tests/test_tvl1_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30) > $O/pytest_batch.log; cat $O/pytest_batch.log
  (timeout 300 python bench.py --workload farneback --steps 3 2>$O/fb.err | tail -1) > $O/farneback_bench.json; cut -c1-1200 $O/farneback_bench.json; tail -3 $O/fb.err ;;
trace_fb)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace_fb -- python $R/bench.py --workload farneback --no-cpu --steps 3 --warmup 1 > $R/$O/trace_fb_bench.log 2>&1
  cd $R
  tail -1 $O/trace_fb_bench.log | cut -c1-900
  for f in $(find $O/trace_fb -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats_farneback.csv; head -16 $f | cut -c1-220; done
  find $O -type f -size +4M -delete ;;
spec_quick)
  for sl in 0 1; do
    (timeout 300 python bench.py --defaults --stop-slack $sl --no-variants --no-cpu --no-secondary --steps 4 --warmup 2 2>$O/spec_sl$sl.err | tail -1) > $O/spec_sl$sl.json
    python - <<PY
import json
try:
    d = json.loads(open('$O/spec_sl$sl.json').read()); print('class defaults stop_slack=$sl', round(d['value'], 1), 'pairs/s', round(d['ms_per_step'], 2), 'ms/step', 'its', d['config']['executed_iterations_per_warp_mean'], 'epe', d['epe_vs_analytic_flow_px'])
except Exception as e: print('spec_quick $sl failed', e); print(open('$O/spec_sl$sl.err').read()[-1500:])
PY
  done ;;
fb_quick)
  (timeout 600 python -m pytest tests/test_farneback.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8) > $O/pytest_fb.log; cat $O/pytest_fb.log
  for t in 1 0; do
    (MIFLOW_FB_TILED=$t timeout 300 python bench.py --workload farneback --no-cpu --steps 3 2>$O/fb$t.err | tail -1) > $O/farneback_bench_tiled$t.json
    python - <<PY
import json
try:
    d = json.loads(open('$O/farneback_bench_tiled$t.json').read()); print('farneback tiled=$t', round(d['value'], 1), 'pairs/s sequential;', d['batched_calc_batch'])
except Exception as e: print('fb_quick $t failed', e); print(open('$O/fb$t.err').read()[-1500:])
PY
  done ;;
ab_planwps)
  for wps in 0 2 1; do
    for mode in fixed defaults; do
      extra=""; [ $mode = defaults ] && extra="--defaults"
      (MIFLOW_TB_WPS=$wps timeout 300 python bench.py $extra --no-variants --no-cpu --no-secondary --steps 6 --warmup 2 2>$O/ab_wps${wps}_$mode.err | tail -1) > $O/ab_wps${wps}_$mode.json
      python - <<PY
import json
try:
    d = json.loads(open('$O/ab_wps${wps}_$mode.json').read()); print('plan wps=$wps $mode', round(d['value'], 1), 'pairs/s', round(d['ms_per_step'], 2), 'ms/step')
except Exception as e: print('ab_planwps $wps $mode failed', e); print(open('$O/ab_wps${wps}_$mode.err').read()[-1500:])
PY
    done
  done ;;
trace_tb5)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace_tb5 -- python $R/bench.py --time-block 5 --no-variants --no-cpu --no-secondary --steps 3 --warmup 1 > $R/$O/trace_tb5_bench.log 2>&1
  cd $R
  python - <<PY
import csv, glob, collections
f = glob.glob('$O/trace_tb5/*/*kernel_trace.csv')[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'tbr' in n:
        key = (n[:44], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000
for k, v in agg.items(): print(k, v[0], round(v[1] / v[0], 1))
PY
  find $O -type f -size +4M -delete ;;
pmc_sq)
  cd /tmp
  rm -rf $R/$O/pmc_sq
  env ${PMC_ENV:-} timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -f csv -d $R/$O/pmc_sq -- python $R/bench.py --lanes 1 --no-variants --no-cpu --no-secondary --steps 2 --warmup 1 > $R/$O/pmc_sq.log 2>&1
  cd $R
  python - <<PY
import csv, glob, collections
fs = glob.glob('$O/pmc_sq/*/*counter_collection.csv')
print(fs)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    w = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, {n: round(x / w, 3) for n, x in v.items()}, 'wave_cycles', w)
PY
  tail -1 $O/pmc_sq.log | cut -c1-200
  for f in $(find $O/pmc_sq -name "*counter_collection.csv" | head -1); do gzip -9 -c $f > $O/pmc_sq_counter_collection.csv.gz; done
  find $O -type f -size +4M -delete ;;
surf_counters)
  # vector-L1 counters + kernel stats of the SURF frame -> profiles/surf_counters.json (tools/surf_counters.py)
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS -f csv -d $R/$O/pmc_surf_l1 -- python $R/bench.py --workload surf --no-cpu --steps 2 --warmup 1 > $R/$O/pmc_surf_l1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace_surf2 -- python $R/bench.py --workload surf --no-cpu --steps 3 --warmup 1 > $R/$O/trace_surf2.log 2>&1
  cd $R
  for f in $(find $O/trace_surf2 -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats_surf.csv; done
  python tools/surf_counters.py $O/pmc_surf_l1 $O/kernel_stats_surf.csv "profiles/$NAME: rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS and --kernel-trace --stats of python bench.py --workload surf --no-cpu (per-launch means)" 2>&1 | tail -10
  cp profiles/surf_counters.json $O/surf_counters.json
  # the raw per-dispatch counters the JSON is derived from, compressed (the file itself is larger than what this script keeps)
  for f in $(find $O/pmc_surf_l1 -name "*counter_collection.csv" | head -1); do gzip -9 -c $f > $O/surf_counter_collection.csv.gz; done
  python tools/surf_timeline.py $O/trace_surf2 > $O/surf_timeline.txt 2>&1
  find $O -type f -size +4M -delete ;;
pmc_secondary)
  cd /tmp
  for wl in ${PMC_WLS:-stereobm farneback surf}; do for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$O/pmc_${wl}_$c -- python $R/bench.py --workload $wl --no-cpu --steps 2 --warmup 1 > $R/$O/pmc_${wl}_$c.log 2>&1
  done; done
  cd $R
  python tools/pmc_secondary.py $O "profiles/$NAME" 2>&1 | tail -5; cp profiles/pmc_traffic.json $O/pmc_traffic.json
  find $O -type f -size +4M -delete ;;
trace_gamma)
  # gamma != 0 on the blocked kernel (round 6): rates of tools/gamma_bench.py, kernel trace and HBM traffic of the N = 10 case
  (timeout 600 python tools/gamma_bench.py 64 4 > $O/gamma_bench.log 2>&1); grep -v amdgpu.ids $O/gamma_bench.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace_gamma -- python $R/tools/gamma_bench.py 64 3 blocked > $R/$O/trace_gamma.log 2>&1
  for f in $(find $R/$O/trace_gamma -name "*kernel_stats.csv" | head -1); do cp $f $R/$O/kernel_stats_gamma.csv; head -6 $f | cut -c1-200; done
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$O/pmc_gamma_$c -- python $R/tools/gamma_bench.py 64 1 blocked > $R/$O/pmc_gamma_$c.log 2>&1
  done
  cd $R
  python - <<PY | tee $O/pmc_gamma_summary.md
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob('$O/pmc_gamma_%s/*/*counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c: continue
            k = r['Kernel_Name'].split('(')[0][-70:]
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:4]:
        print(f"| {c} | {k} | launches {n} | mean per launch {v / max(n, 1):.1f} (counter units: KB; FETCH_SIZE x 2 on gfx950) |")
PY
  find $O -type f -size +4M -delete ;;
trace_secondary)
  # kernel traces of the secondary workloads on the final tree (VERDICT r03 hygiene item)
  cd /tmp
  for wl in stereobm farneback surf; do
    timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/trace_$wl -- python $R/bench.py --workload $wl --no-cpu --steps 3 --warmup 1 > $R/$O/trace_${wl}_bench.log 2>&1
    for f in $(find $R/$O/trace_$wl -name "*kernel_stats.csv" | head -1); do cp $f $R/$O/kernel_stats_$wl.csv; echo "== $wl"; head -8 $f | cut -c1-170; done
  done
  cd $R
  find $O -type f -size +4M -delete ;;
spec_trace)
  (timeout 300 python tools/spec_trace.py --pairs 4 2>&1 | tail -120) > $O/spec_trace.log; head -70 $O/spec_trace.log ;;
jw2)
  J=${JW_B:-2}
  for jw in 0 $J; do (MIFLOW_TB_JW=$jw timeout 600 python tools/jw_check.py > $O/jw_digest_$jw.txt 2>$O/jw_$jw.err); tail -3 $O/jw_digest_$jw.txt; tail -2 $O/jw_$jw.err; done
  if diff <(grep -v "^#" $O/jw_digest_0.txt) <(grep -v "^#" $O/jw_digest_$J.txt) > $O/jw_diff.txt; then echo "JW$J DIGESTS EQUAL"; else echo "JW$J DIGESTS DIFFER"; head -20 $O/jw_diff.txt; fi ;;
surf_bench)
  (timeout 300 python bench.py --workload surf --no-cpu --steps 5 2>$O/surf.err | tail -1) > $O/surf_bench.json; cut -c1-1800 $O/surf_bench.json; tail -3 $O/surf.err ;;
test_one)
  (timeout 600 python -m pytest "tests/test_baseline_sizes.py" -m gpu -q -x -p no:cacheprovider -k "two_lanes or speculative or slack or class_defaults" 2>&1 | tail -30) > $O/pytest_one.log; cat $O/pytest_one.log ;;
esac
done
