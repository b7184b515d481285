#!/usr/bin/env bash
# First session on a node with >= 2 MI355X (the reference's own concurrency evidence is
# cudaoptflow/test/test_optflow.cpp:468-527).  No such node has been available to the build: this script runs, in order, everything that
# has only ever executed on ONE device, and prints what a scaling table needs.  Every step runs under `timeout`, writes its log under
# $OUT, and a failing step does not stop the later ones.
#
#   usage: bash tools/multi_gpu_first_run.sh [OUT_DIR]        (from the repository root; N = all visible GPUs, capped at 8)
#
#   1. the two GPU tests that skip on a one-GPU box (distinct devices through mi_tvl1_multi; bench.py --gpus 2 over RCCL)
#   2. mi_tvl1_multi over all devices with RCCL links, then with MIFLOW_MULTI_RCCL=0 (peer copies): bit-identity against calc_batch,
#      transport() / transportWhy(), pairs per second
#   3. bench.py --gpus 1 | 2 | 4 | 8, each with the scatter / gather leg and with MIFLOW_BENCH_EXCHANGE=0
#   4. the table: value per N, efficiency = value_N / (N * value_1), per-rank rates
set -u
OUT=${1:-gpurun_out/multi_gpu_first_run}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export MASTER_ADDR=127.0.0.1
NGPU=$(python - <<'PY'
import torch
print(min(torch.cuda.device_count(), 8))
PY
)
echo "[multi] $NGPU visible GPUs" | tee "$OUT/summary.txt"
if [ "$NGPU" -lt 2 ]; then echo "[multi] needs >= 2 GPUs: nothing to do" | tee -a "$OUT/summary.txt"; exit 0; fi

step() {   # step <name> <seconds> <command...>
  local name=$1 lim=$2; shift 2
  echo "[multi] ---- $name" | tee -a "$OUT/summary.txt"
  timeout "$lim" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "[multi] $name rc=$rc: $(tail -n 1 "$OUT/$name.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
  return 0
}

# 1. the tests a one-GPU box skips
step tests_distinct_devices 1200 python -m pytest tests/test_baseline_sizes.py -q -x -k "distinct_devices or two_ranks_over_rccl"
step tests_parallel_cpu 600 python -m pytest tests/test_parallel_cpu.py -q -x

# 2. the C++ entry over all devices: RCCL links, then peer copies
cat > "$OUT/multi_entry.py" <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from opencv_contrib_amd import cuda, synth
nd = min(torch.cuda.device_count(), 8)
dev = torch.device("cuda", 0)
n = 16 * nd
base = [synth.flow_pair(1080, 1920, seed=1234 + k)[:2] for k in range(4)]
I0s = [torch.from_numpy(base[k % 4][0]).to(dev) for k in range(n)]
I1s = [torch.from_numpy(base[k % 4][1]).to(dev) for k in range(n)]
alg = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0)
ref = alg.calc_batch(I0s[:32], I1s[:32]); torch.cuda.synchronize()
for devices in (list(range(nd)), list(range(nd - 1, -1, -1))):
    multi = cuda.TVL1MultiDevice(alg, devices=devices, chunk=16)
    out = multi.calc_batch(I0s, I1s)
    same = all(torch.equal(out[k], ref[k % 32]) for k in range(n))   # (the inputs repeat with period 4)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        multi.calc_batch(I0s, I1s, out)
    el = time.perf_counter() - t0
    print(f"devices {devices}: transport (rccl, peer-copy) = {multi.transport()} why='{multi.transportWhy()}' "
          f"all flows identical to calc_batch: {same}; {n * reps / el:.0f} pairs/s over {nd} GPUs", flush=True)
    del multi
PY
step multi_entry_rccl 900 python "$OUT/multi_entry.py"
step multi_entry_peer_copies 900 env MIFLOW_MULTI_RCCL=0 python "$OUT/multi_entry.py"

# 3. the bench at 1 / 2 / 4 / 8 ranks, with and without the exchange leg
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  step "bench_gpus${N}" 1500 python bench.py --gpus "$N" --steps 10 --warmup 2 --no-variants --no-secondary --no-cpu
  cp -f bench_full.json "$OUT/bench_full_gpus${N}.json" 2>/dev/null
  if [ "$N" -gt 1 ]; then
    step "bench_gpus${N}_no_exchange" 1500 env MIFLOW_BENCH_EXCHANGE=0 python bench.py --gpus "$N" --steps 10 --warmup 2 --no-variants --no-secondary --no-cpu
  fi
done

# 4. the table
python - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import json, os, sys
out = sys.argv[1]
def line(p):
    try:
        ls = [l for l in open(p).read().splitlines() if l.startswith("{")]
        return json.loads(ls[-1]) if ls else None
    except OSError:
        return None
base = None
print("N  pairs/s      efficiency  ms/step  with scatter/gather   ranks on distinct devices")
for N in (1, 2, 4, 8):
    d = line(os.path.join(out, f"bench_gpus{N}.log"))
    if not d:
        continue
    if N == 1:
        base = d["value"]
    eff = d["value"] / (N * base) if base else float("nan")
    wsg = (d.get("with_scatter_gather") or {})
    rr = d.get("rccl_ranks") or {}
    print(f"{N}  {d['value']:10.1f}  {eff:9.3f}  {d['ms_per_step']:7.2f}  {wsg.get('pairs_per_s', wsg.get('error', '-'))}  {rr.get('distinct_devices')}")
    f = os.path.join(out, f"bench_full_gpus{N}.json")
    if os.path.exists(f):
        full = json.load(open(f))
        per = full.get("per_rank_pairs_per_s") or full.get("per_rank") or None
        if per:
            print("   per-rank:", per)
PY
echo "[multi] logs under $OUT" | tee -a "$OUT/summary.txt"
