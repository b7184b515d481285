#!/usr/bin/env python
"""bench.py -- frame-pairs/s of dense Dual TV-L1 flow at 1080p on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (mi_tvl1_calc_batch through the C-ABI) over one batch of
`--batch` synthetic 1920x1080 CV_32FC1 pairs already resident in HBM.  Workload at N=1 is
BASELINE.json configs[1] ("DualTVL1 dense flow, 1920x1080, 1xMI355X"); with --gpus N every rank
processes its own batch (independent pairs, no data-path collective: weak scaling, configs[4]); the same job
with the pairs resident on GPU 0 only (RCCL scatter / gather inside the timed region, overlapped with compute) is
reported next to it as "with_scatter_gather" (MIFLOW_BENCH_EXCHANGE=0 skips that leg).

Primary parameter set: the reference accuracy-test setting iterations=10 with epsilon=0
(fixed work: cudaoptflow/test/test_optflow.cpp:450; CPU equivalent median=1, inner=1, outer=10)
unless --defaults is given (class defaults: 300 iterations, epsilon 0.01, data-dependent exit).
The JSON line carries N and the per-pair algorithmic bytes so the number can be read against
BASELINE.md section 2.  Extra parameter sets are reported under "variants".

Rank 0 prints ONE compact JSON line (<= 4 KB, strict JSON: `compact_line`) as the only line on stdout; the long form
(variants, secondary workloads, per-level tables, notes) goes to bench_full.json and to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# MI355X_MICROARCH.md: 157.3 TFLOP/s f32 vector peak = 256 CUs x 4 SIMD-32 x 2.4 GHz x 2 flop per fma lane-instruction
VALU_PEAK_TLIPS = 157.3 / 2
# vector-memory data path: 64 B/clk per CU (MI355X_MICROARCH.md: a dwordx4 wave load = 16 clk) x 256 CUs x 2.4 GHz
VMEM_PATH_PEAK_GBS = 64 * 256 * 2.4


def sbm_valu_per_pxd():
    """VALU instructions of k_block_match's row loop per (output pixel, disparity) of a tile row: static count of the R = 7
    instantiation (tools/static_mix.py sbm -> profiles/static_mix_sbm.json): column-sum slide (v_sub_u32_sdwa + v_mad_i32_i24, both
    rows), window slide, score packing and the transposed wave max-reduction with its selects -- 29.4, not the ~10 arithmetic
    operations SURVEY 8d config 3 estimated (which the round-1 figure used)."""
    try:
        return float(json.load(open(os.path.join(ROOT, "profiles", "static_mix_sbm.json")))["valu_per_output_pixel_and_disparity"])
    except Exception:
        return 17.375


def sbm_warmup_ratio():
    """VALU count of a band's warm-up row (the first 2R rows only add to the column sums) relative to a full row."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "static_mix_sbm.json")))
        return float(d["warmup_row_valu"]) / float(d["row_loop_valu"])
    except Exception:
        return 157.0 / 834.0


def sbm_band_rows(rows, cols, ndisp, R, pairs):
    """Rows per band of block_match_impl (stereobm_kernels.hip): every band slides 2R rows before its first output row."""
    xt = -(-(cols - ndisp - 2 * R) // 48)
    bands = -(-5120 // (xt * -(-ndisp // 64) * pairs))
    rb = -(-(rows - 2 * R) // max(bands, 1))
    return min(max(rb, 2 * R + 2), 48)


def level_pixels(w, h, nscales=5, step=0.8):
    import numpy as np
    px = []
    for s in range(nscales):
        if s:
            w, h = int(np.rint(w * step)), int(np.rint(h * step))
            if w < 16 or h < 16:
                break
        px.append(w * h)
    return px


def algo_bytes_per_pair(w, h, warps, iters_per_warp, nscales=5, step=0.8):
    """sum_levels px * (12 + warps*(44 + 64*N))."""
    return float(sum(p * (12 + warps * (44 + 64 * iters_per_warp)) for p in level_pixels(w, h, nscales, step)))


WARP_VALU_PER_PX = 204.0   # static count of the interior path of k_warp6<CPU_REF, 32, ., exact sums>: 52 (map, phase weights) + 139 (window; 169 before round 3 hoisted the 32 multiplies by 0.5 out of the tap products) + 13 (grad, rho_c)


def tbr_jw():
    """MIFLOW_TB_JW as the library reads it: 2 (default) = joined waves, barrier form; 1 = joined waves, tags; 0 = independent waves."""
    try:
        return int(os.environ.get("MIFLOW_TB_JW", "2"))
    except ValueError:
        return 2


def tbr_band_rows(w, h, pairs, T=10, PF=2, plan_wps=3, simds=1024, jw=None):
    """Band height the planner of tvl1_tbr_kernels.hip (plan_band_rows) picks for the T = 10, 1 px/lane kernel: every wave streams
    rows + 2 T rows, so (rows + 2 T) / rows of the iteration kernel's work is band-halo recomputation.  jw: a workgroup's four waves
    sit side by side on one band of a 256-column strip (the default) instead of on four bands of a 64-column strip."""
    jw = tbr_jw() if jw is None else jw
    P = T + 1 + PF
    nw = 8 if jw == 3 else 4
    M, LW = T, (64 * nw if jw else 64)
    strips = 1 if w <= LW - M else 1 + -(-(w - (LW - M)) // (LW - 2 * M))
    per_band, cap = strips * pairs * (nw if jw else 1), simds * plan_wps
    best = None
    for nb in range(1, h + 1):
        R = -(-h // nb)
        if R < 8 and nb > 1:
            break
        cap_nb = simds * nb if not jw and nb < 4 and nb < plan_wps else cap
        rounds = -(-(per_band * nb) // cap_nb)
        steps = -(-(R + 2 * T) // P) * P
        if best is None or rounds * steps < best[0]:
            best = (rounds * steps, nb)
    return -(-h // best[1])


def _gen_pair(a):
    from opencv_contrib_amd import synth
    h, w, seed, kw = a
    return synth.flow_pair(h, w, seed=seed, **kw)


def gen_base_pairs(n, h, w, **kw):
    """n DISTINCT synthetic pairs (seeds 1234 ...), generated in worker processes (numpy / scipy only; call it before the first CUDA
    call of the process: the workers are forked)."""
    import multiprocessing as mp
    jobs = [(h, w, 1234 + i, kw) for i in range(n)]
    # under rocprofv3 (LD_PRELOAD of its tool library) forked workers never return (r03n: three profiled runs sat in the pool until
    # their timeouts): generate in-process there
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if n <= 1 or profiled or os.environ.get("MIFLOW_BENCH_NO_FORK"):
        return [_gen_pair(j) for j in jobs]
    try:
        with mp.get_context("fork").Pool(min(n, max(1, (os.cpu_count() or 2) // 2))) as pool:
            return pool.map(_gen_pair, jobs)
    except Exception:
        return [_gen_pair(j) for j in jobs]


def make_inputs(n, h, w, dev, distinct=16, base=None, **kw):
    """n pairs on `dev` built from min(n, distinct) distinct pairs (the iteration histogram of the class-default variant
    must not be that of 4 images); pairs beyond the distinct ones repeat them."""
    import torch
    if base is None:
        base = gen_base_pairs(min(n, distinct), h, w, **kw)
    I0 = torch.stack([torch.from_numpy(base[i % len(base)][0]) for i in range(n)]).to(dev)
    I1 = torch.stack([torch.from_numpy(base[i % len(base)][1]) for i in range(n)]).to(dev)
    return I0, I1, base


def synth_sequence(n, h, w, dev, seed=4321):
    """n consecutive frames (CV_32FC1 in [0, 1], on `dev`) of one synthetic sequence: a band-limited texture drifting along the smooth
    field of synth.flow_field -- frame k samples the texture at p - k x F(p) (bicubic) -- so consecutive pairs have a smooth, slowly
    varying flow of a few pixels, like neighbouring frames of a video."""
    import numpy as np
    import torch
    from opencv_contrib_amd import synth
    sc = max(1.0, w / 640.0)
    tex = torch.from_numpy((synth.texture(h, w, seed, 2.0 * sc) / 255.0).astype(np.float32)).to(dev)[None, None]
    u, v = synth.flow_field(h, w, 0.5 * sc)
    u, v = torch.from_numpy(u.astype(np.float32)).to(dev), torch.from_numpy(v.astype(np.float32)).to(dev)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    out = []
    for k in range(n):
        gx = (xs - k * u) / (w - 1) * 2 - 1
        gy = (ys - k * v) / (h - 1) * 2 - 1
        f = torch.nn.functional.grid_sample(tex, torch.stack([gx, gy], -1)[None], mode="bicubic", padding_mode="reflection", align_corners=True)
        out.append(f[0, 0].clamp(0, 1).contiguous())
    return out


def omp_set_threads(n):
    """Thread count of the OpenMP runtime the oracle libraries are linked against (libgomp)."""
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except Exception:
        return False


def physical_cores():
    try:
        seen = set()
        for c in os.listdir("/sys/devices/system/cpu"):
            if c.startswith("cpu") and c[3:].isdigit():
                f = f"/sys/devices/system/cpu/{c}/topology/thread_siblings_list"
                if os.path.exists(f):
                    seen.add(open(f).read().strip())
        return max(1, min(len(seen) or (os.cpu_count() or 1), len(os.sched_getaffinity(0))))
    except Exception:
        return os.cpu_count() or 1


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited or unknown.  On the GPU boxes of this
    pool the quota is 16 of the host's 128 cores: more threads than that are throttled, which is why the thread sweep of the CPU baseline
    turns over at 16 (round 6; VERDICT r05 weak item 9 read it as the stub scheduler's doing)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def timed_cpu_baseline(cpu_calc, base, budget_s=12.0, hard_s=25.0, with_one_core=True):
    """CPU-baseline protocol of BASELINE.md section 3 on a bounded sample: thread count capped at the PHYSICAL cores (a short sweep
    picks the fastest of {physical, 64, 32, 16, 8}: stripes of a parallel_for_ over-subscribe badly on a 256-thread host), one warm-up
    at the chosen count, then >= 5 timed repetitions (up to 9 inside `budget_s`; never past `hard_s` once 3 are in) over the distinct
    pairs of `base`; median and min; plus pair 0 on ONE core.  cpu_calc(I0, I1) -> flow."""
    import numpy as np

    def one(a, b):
        t_ = time.perf_counter()
        r_ = cpu_calc(a, b)
        return time.perf_counter() - t_, r_

    phys = physical_cores()
    b0_ = base[0]
    sweep, ref0 = {}, None
    for nt in sorted({phys, min(phys, 64), min(phys, 32), min(phys, 16), min(phys, 8)}, reverse=True):
        if not omp_set_threads(nt):
            nt = os.cpu_count() or 1
        dt, ref0 = one(b0_[0], b0_[1])      # the first one doubles as a warm-up (first touch of the buffers, thread pool start)
        sweep[nt] = min(dt, sweep.get(nt, dt))
        if len(sweep) > 1 and dt > 2.0 * min(sweep.values()):
            break
    best_nt = min(sweep, key=sweep.get)
    omp_set_threads(best_nt)
    one(b0_[0], b0_[1])                     # warm-up at the chosen thread count
    times, tstart = [], time.perf_counter()
    while True:
        b_ = base[len(times) % len(base)]
        times.append(one(b_[0], b_[1])[0])
        el_ = time.perf_counter() - tstart
        if len(times) >= 9 or (len(times) >= 5 and el_ > budget_s) or (len(times) >= 3 and el_ > hard_s):
            break
    one_core = None
    if with_one_core:
        try:
            if omp_set_threads(1):
                dt1, _ = one(b0_[0], b0_[1])
                one_core = {"value": 1.0 / dt1, "unit": "pairs/s", "cores": 1, "sample": f"1 pair, 1 repetition, {dt1:.1f} s"}
        finally:
            omp_set_threads(best_nt)
    return {"ref0": ref0, "times": times, "median_s": float(np.median(times)), "threads": best_nt, "physical_cores": phys,
            "sweep": sweep, "one_core": one_core}


class PowerPoll:
    """rocm-smi polled from a thread while a leg runs: shader clock and socket power.  The TV-L1 step runs MI355X at its POWER limit
    (about 1350 W; the shader clock settles near 2.05 GHz instead of 2.4), so the clock and the power belong next to any rate or
    roofline fraction quoted against the 2.4 GHz peaks."""

    def __init__(self, period=0.12):
        import threading
        self.samples, self._stop, self.period = [], False, period
        self.th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                p_ = re.search(r"Power \(W\): ([\d.]+)", o)
                if c and p_:
                    self.samples.append((time.perf_counter(), int(c.group(1)), float(p_.group(1))))
            except Exception:
                return
            time.sleep(self.period)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self.th.join(timeout=6)

    def mean(self, t0, t1):
        s_ = [x for x in self.samples if t0 + 0.35 * (t1 - t0) <= x[0] <= t1]   # the settled part of the leg
        if len(s_) < 2:
            return None
        return {"sclk_MHz": sum(x[1] for x in s_) / len(s_), "socket_power_W": sum(x[2] for x in s_) / len(s_), "samples": len(s_)}


def power_leg(fn, units_per_call, seconds=2.5):
    """Runs fn() back to back for `seconds` with the clock / power poll beside it; returns the settled means and the leg's own rate."""
    import torch
    try:
        with PowerPoll() as poll:
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < seconds:
                fn()
                torch.cuda.synchronize()
                n += 1
            t1 = time.perf_counter()
            m = poll.mean(t0, t1)
        if not m:
            return None
        m.update({"rate_during_leg": n * units_per_call / (t1 - t0), "leg_s": t1 - t0})
        return m
    except Exception as e:
        return {"error": repr(e)[:200]}


def time_steps(alg, I0, I1, flows, steps, warmup, dist):
    """I0 / I1: one batch, or a LIST of batches used in turn (step k computes batch k mod len -- the variant that measures the
    convergence-checked path with its block-length history coming from OTHER pairs than the ones being computed)."""
    import torch
    if isinstance(I0, (list, tuple)):
        seq0, seq1 = list(I0), list(I1)
        for k in range(warmup):
            alg.calc_batch(seq0[k % len(seq0)], seq1[k % len(seq0)], flows)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            alg.calc_batch(seq0[(warmup + k) % len(seq0)], seq1[(warmup + k) % len(seq0)], flows)
        torch.cuda.synchronize()
        time_steps.local_s = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0
    for _ in range(warmup):
        alg.calc_batch(I0, I1, flows)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        alg.calc_batch(I0, I1, flows)
    torch.cuda.synchronize()
    time_steps.local_s = time.perf_counter() - t0   # this rank's own time, before it waits for the others
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0


def bench_exchange(args, parallel, dist, rank, world, dev, I0, I1, flows_ref):
    """Batched-frames mode with the pairs resident on GPU 0 only (BASELINE north_star / SURVEY 8e): every step scatters the
    batches over RCCL, computes, and gathers the flows back to GPU 0; scatter of step k+1 and gather of step k-1 overlap the
    compute of step k (opencv_contrib_amd/parallel.py).  Reported next to `value`, never as `value` (which is measured with the
    inputs already resident on every GPU)."""
    import torch
    from opencv_contrib_amd import cuda
    x = torch.stack([I0, I1], 1)                                   # (B, 2, H, W)
    # BASELINE configs[4]: 64 pairs per GPU (512 over 8) -- the timed batch tiled with row shifts so that the 64 pairs are distinct
    per_gpu = max(int(os.environ.get("MIFLOW_BENCH_EXCHANGE_PAIRS", "64")), x.shape[0])
    reps = (per_gpu + x.shape[0] - 1) // x.shape[0]
    x = torch.cat([torch.roll(x, 8 * k, 2) for k in range(reps)], 0)[:per_gpu].contiguous()
    flows_ref = torch.empty((x.shape[0],) + tuple(flows_ref.shape[1:]), dtype=flows_ref.dtype, device=flows_ref.device)
    # rank 0 holds `world` DISTINCT shards (the batch rolled by 16 r columns): every rank receives and computes different pairs
    parts = [torch.roll(x, 16 * r, 3).contiguous() for r in range(world)] if rank == 0 else None
    local_in = [torch.empty_like(x) for _ in range(2)]
    local_out = [torch.empty_like(flows_ref) for _ in range(2)]
    root_out = [[torch.empty_like(flows_ref) for _ in range(world)] for _ in range(2)] if rank == 0 else None
    alg = cuda.OpticalFlowDual_TVL1.create(iterations=args.iterations, epsilon=args.epsilon,
                                           exactMath=True if args.exact_math else None, timeBlock=args.time_block)

    def compute(inp, out):
        alg.calc_batch(inp[:, 0], inp[:, 1], out)

    sync = torch.cuda.synchronize
    parallel.run_exchange_pipeline(dist, rank, world, parts, local_in, local_out, root_out, compute, max(1, args.warmup), sync)
    el = parallel.run_exchange_pipeline(dist, rank, world, parts, local_in, local_out, root_out, compute, args.steps, sync)
    el = parallel.max_over_ranks(dist, el, dev)
    B = x.shape[0]
    res = {"value": B * world * args.steps / el, "unit": "pairs/s", "ms_per_step": 1e3 * el / args.steps, "pairs_per_gpu": B,
           "exchange_GB_per_step": (world - 1) * (x.numel() + flows_ref.numel()) * 4 / 1e9,
           "note": "inputs on GPU 0 only (one distinct shard per rank): scatter (grouped RCCL send/recv) -> calc_batch -> gather of the "
                   "flows to GPU 0 inside the timed region, double buffered"}
    if rank == 0:
        last = root_out[(args.steps - 1) & 1]
        # the kernels are deterministic: the flows gathered from rank r must equal rank 0's own computation of shard r
        ok = True
        chk = torch.empty_like(flows_ref)
        for r in range(world):
            alg.calc_batch(parts[r][:, 0], parts[r][:, 1], chk)
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(last[r], chk))
        res["gathered_flows_identical"] = ok
    return res


def bench_stereobm(args):
    """BASELINE configs[2]: StereoBM 1920x1080, numDisparities=128, blockSize=15 (secondary workload;
    the headline metric stays TV-L1).  One step = `--batch` stereo pairs through mi_stereobm_compute."""
    import numpy as np
    import torch
    from opencv_contrib_amd import cuda, synth
    dev = torch.device("cuda", 0)
    W, H, B = args.width, args.height, args.batch
    nd, bs = args.ndisp, args.block_size
    left, right, _ = synth.stereo_pair(H, W, seed=42, max_disp=70)
    L = [torch.from_numpy(left).to(dev) for _ in range(B)]
    Rr = [torch.from_numpy(right).to(dev) for _ in range(B)]
    D = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in range(B)]
    bm = cuda.createStereoBM(nd, bs)
    for _ in range(args.warmup):
        for i in range(B):
            bm.compute(L[i], Rr[i], D[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        for i in range(B):
            bm.compute(L[i], Rr[i], D[i])
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n = B * args.steps
    # the batched entry: block matching of the B pairs in one launch (taller row bands), results identical to compute()
    Db = [torch.empty_like(D[0]) for _ in range(B)]
    bm.compute_batch(L, Rr, Db)
    torch.cuda.synchronize()
    same = all(bool(torch.equal(Db[i], D[i])) for i in range(B))
    tb = time.perf_counter()
    for _ in range(args.steps):
        bm.compute_batch(L, Rr, Db)
    torch.cuda.synchronize()
    elb = time.perf_counter() - tb
    batched = {"pairs_per_s": n / elb, "batch": B, "equals_single_compute": same}
    R = bs // 2
    pxd = float((W - nd - 2 * R) * (H - 2 * R)) * nd
    algo_bytes = 3.0 * W * H   # read left + right, write disparity (u8)
    vpd, R_ = sbm_valu_per_pxd(), bs // 2
    rb1, rbb = sbm_band_rows(H, W, nd, R_, 1), sbm_band_rows(H, W, nd, R_, B)
    halo_seq, halo_b = 1.0 + 2.0 * R_ * sbm_warmup_ratio() / rb1, 1.0 + 2.0 * R_ * sbm_warmup_ratio() / rbb
    out = {"metric": "frames/sec StereoBM @1080p", "value": n / el, "unit": "pairs/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"StereoBM {W}x{H} numDisparities={nd} blockSize={bs} (BASELINE configs[2]), {B} pairs/step",
                      "texture_threshold": 3, "uniqueness_ratio": 0},
           "pixel_disparities_per_s": pxd * n / el,
           "batched_compute_batch": batched, "batched_pixel_disparities_per_s": pxd * n / elb,
           # not HBM-bound (6.2 MB/pair); the work is (pixel, disparity) cost updates on the integer VALU.
           # achieved = EXECUTED lane-instructions per second: static VALU count of the row loop per (output pixel, disparity)
           # x (1 + 2R x warm-up row cost / rb): the 2R warm-up rows of a band only add to the column sums (rb = rows per band of
           # the launch plan), against the issue peak
           "roofline": {"bound": "valu_issue", "achieved": pxd * n / el * vpd * halo_seq / 1e12, "peak": VALU_PEAK_TLIPS,
                        "unit": "T lane-instr/s", "frac": pxd * n / el * vpd * halo_seq / 1e12 / VALU_PEAK_TLIPS,
                        "batched_frac": pxd * n / elb * vpd * halo_b / 1e12 / VALU_PEAK_TLIPS,
                        "valu_per_pixel_disparity": vpd, "executed_per_useful_row_work": {"sequential": halo_seq, "batched": halo_b},
                        "useful_frac_batched": pxd * n / elb * vpd / 1e12 / VALU_PEAK_TLIPS,
                        "hbm_algorithmic_GBps": algo_bytes * n / el / 1e9, "hbm_frac": algo_bytes * n / el / 1e9 / HBM_PEAK_GBS,
                        # measured per launch of ONE pair (compute()) and per pair of a batched launch, kept apart since round 4
                        # (the r07v / r09z figure of 71 MB was the mean over both kinds of launch, and 25 + 12 MB of the 37 MB a
                        # pair really moved were the R rows fetched once per XCD and the unread winners' SSD plane: both gone)
                        "traffic": pmc_traffic("stereobm", sub="one_pair_launch")[0], "traffic_kernel": "k_block_match, bytes per launch of one pair",
                        "traffic_per_pair_batched": pmc_traffic("stereobm", sub="batched_launch")[0],
                        "traffic_algorithmic_bytes_per_pair": algo_bytes,
                        "traffic_source": pmc_traffic("stereobm")[1]}}
    # The 2-cycle issue peak holds for a short list of gfx950 VALU operations only (fma / mul / add / sub f32, add / sub u32, and / or /
    # xor, right shift, mov); SDWA, DPP, selects, min / max, 24-bit multiplies, left shifts, conversions take 4.2 cycles per wave
    # instruction (round 4: tools/ubench/issue_mix.hip, profiles/valu_rates_gfx950.json).  rate_weighted: the row loop's SIMD cycles if
    # every instruction issued at ITS measured rate against the cycles the kernel's share of a pair actually takes -- what the kernel can
    # still gain without changing its instruction mix.
    try:
        mixd = json.load(open(os.path.join(ROOT, "profiles", "static_mix_sbm.json")))
        cyc = float(mixd["rate_weighted_cycles_per_output_pixel_and_disparity_lane"])   # SIMD cycles per (pixel, disparity) lane-unit x 64
        simds, clk = 1024.0, 2.39e9   # StereoBM is not power-limited: the clock stays at 2.39 GHz (profiles/r08/pmc_sq_block_match.txt, r08j poll)
        peak_pxd = simds * 64.0 * clk / cyc
        out["roofline"]["rate_weighted"] = {"cycles_per_pixel_disparity_per_lane": cyc, "mix": mixd.get("rate_weighted"),
                                            "peak_pixel_disparities_per_s": peak_pxd, "clock_GHz": clk / 1e9,
                                            "frac_batched_executed": pxd * n / elb * halo_b / peak_pxd,
                                            "frac_sequential_executed": pxd * n / el * halo_seq / peak_pxd}
    except Exception as e:
        out["roofline"]["rate_weighted"] = {"error": repr(e)[:200]}
    if batched["equals_single_compute"]:
        # `value` = the batched entry (block matching of the B pairs in one launch); the sequential compute() loop stays next to it
        out["sequential_compute_pairs_per_s"] = out["value"]
        out["value"] = batched["pairs_per_s"]
        out["ms_per_step"] = 1e3 * B / batched["pairs_per_s"]
        out["config"]["workload"] += "; value = compute_batch"
        out["roofline"]["sequential_frac"] = out["roofline"]["frac"]
        out["roofline"]["achieved"] = pxd * n / elb * vpd * halo_b / 1e12
        out["roofline"]["frac"] = out["roofline"]["batched_frac"]
    # post-filter of the stereo pipeline: DisparityBilateralFilter(ndisp, radius 3, 1 iteration) on the maps above
    dbf = cuda.createDisparityBilateralFilter(nd, 3, 1)
    F = [torch.empty_like(D[0]) for _ in range(B)]
    for i in range(B):
        dbf.apply(D[i], L[i], F[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for i in range(B):
            dbf.apply(D[i], L[i], F[i])
    torch.cuda.synchronize()
    eld = time.perf_counter() - t0
    out["disparity_bilateral_filter_maps_per_s"] = n / eld
    out["disparity_bilateral_filter_refined_fraction"] = float((F[0] != D[0]).float().mean())
    # semi-global matching on the same pairs: StereoSGM(0, ndisp, 10, 120, 5, MODE_HH4), CV_16SC1 output
    sgm = cuda.createStereoSGM(0, nd, 10, 120, 5, cuda.StereoSGM.MODE_HH4)
    S = torch.empty((H, W), dtype=torch.int16, device=dev)
    sgm.compute(L[0], Rr[0], S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nsg = max(2, min(B, 4)) * args.steps
    for k in range(nsg):
        sgm.compute(L[k % B], Rr[k % B], S)
    torch.cuda.synchronize()
    els = time.perf_counter() - t0
    out["stereosgm_hh4_pairs_per_s"] = nsg / els
    out["stereosgm_valid_fraction"] = float((S >= 0).float().mean())
    if not args.no_cpu:
        from oracle import oracle as O
        t0 = time.perf_counter()
        dref = O.sbm_compute(left, right, O.sbm_params(num_disparities=nd, block_size=bs))
        ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / ct, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"1 pair {W}x{H}, {ct:.1f} s wall, oracle/stereobm_ref.c (OpenMP rows)"}
        t0 = time.perf_counter()
        O.dbf_apply(dref, left, O.dbf_params(ndisp=nd, radius=3, iters=1))
        out["cpu_baseline"]["disparity_bilateral_filter_maps_per_s"] = 1.0 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        O.sgm_compute(left, right, O.sgm_params(num_disparities=nd))
        out["cpu_baseline"]["stereosgm_hh4_pairs_per_s"] = 1.0 / (time.perf_counter() - t0)
    return out


def bench_farneback(args):
    """BASELINE configs[0]: calcOpticalFlowFarneback on one 640x480 synthetic pair, class defaults (the
    reference's own CPU-runnable case) -- GPU path (mi_farneback_calc) next to the CPU restatement."""
    import numpy as np
    import torch
    from opencv_contrib_amd import cuda, synth
    dev = torch.device("cuda", 0)
    W, H = args.width, args.height
    I0, I1, gt = synth.flow_pair(H, W, seed=1234, dtype="u8")
    t0_, t1_ = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
    flow = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
    alg = cuda.FarnebackOpticalFlow.create()
    n = max(args.batch, 1)
    for _ in range(args.warmup * n):
        alg.calc(t0_, t1_, flow)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps * n):
        alg.calc(t0_, t1_, flow)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    f = flow.cpu().numpy()
    # batched-frames mode: nb distinct-buffer copies of the pair per mi_farneback_calc_batch (blockIdx.z = pair in every kernel):
    # a 640 x 480 pair alone is ~70 launches of 5-50 us, the batch shares them
    batched = None
    try:
        nb = 32
        b0, b1 = [t0_.clone() for _ in range(nb)], [t1_.clone() for _ in range(nb)]
        bflows = torch.empty((nb, H, W, 2), dtype=torch.float32, device=dev)
        balg = cuda.FarnebackOpticalFlow.create()
        balg.calc_batch(b0, b1, bflows)
        torch.cuda.synchronize()
        reps = max(2, args.steps)
        tb = time.perf_counter()
        for _ in range(reps):
            balg.calc_batch(b0, b1, bflows)
        torch.cuda.synchronize()
        batched = {"pairs_per_s": nb * reps / (time.perf_counter() - tb), "batch": nb,
                   "equals_single_calc": bool(torch.equal(bflows[nb - 1], flow))}
        del b0, b1, bflows, balg
    except Exception as e:
        batched = {"error": repr(e)[:200]}
    # algorithmic bytes of the fused formulation: per level 2x polyexp 24 + updateMatrices 68 + iters*(M 20 + R0 20 + R1 20 + flow 8 + M' 20)
    px = 0
    w_, h_ = W, H
    lv = []
    scale = 1.0
    for k in range(6):
        if k and (W * scale < 32 or H * scale < 32):
            break
        lv.append(int(np.rint(W * scale)) * int(np.rint(H * scale)))
        scale *= 0.5
    lv = lv[:6]
    algo = float(sum(p * (2 * 24 + 68 + 10 * 88) for p in lv))
    out = {"metric": "frame-pairs/sec Farneback flow", "value": args.steps * n / el, "unit": "pairs/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"FarnebackOpticalFlow {W}x{H} CV_8UC1, numLevels 5, pyrScale .5, winSize 13, numIters 10, polyN 5 "
                                  f"(BASELINE configs[0]), {n} sequential calc()/step"},
           "epe_vs_analytic_flow_px": float(synth.epe(f[40:-40, 40:-40], gt[40:-40, 40:-40])),
           "batched_calc_batch": batched,
           "roofline": {"bound": "hbm", "achieved": algo * args.steps * n / el / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo * args.steps * n / el / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("farneback_iterate_level0_batch")[0],
                        "traffic_kernel": (lambda q: f"k_iterate_t, finest level of the batched calc: bytes per launch ({q} pairs x 640 x 480 px; "
                                                     f"algorithmic 88 B/px = {q * 640 * 480 * 88 / 1e6:.0f} MB)")(pmc_pairs("farneback_iterate_level0_batch", 32)),
                        "traffic_source": pmc_traffic("farneback_iterate_level0_batch")[1],
                        "note": "sequential calc() of ONE small pair: launch-latency bound (about 70 launches of 5-50 us); bytes = "
                                "fused-iteration accounting"}}
    if batched and "pairs_per_s" in batched:
        # `value` = the batched entry (mi_farneback_calc_batch, the product's throughput path, as mi_tvl1_calc_batch is for the
        # headline); the one-pair-per-calc() rate of the reference's own calling pattern stays next to it
        out["sequential_calc_pairs_per_s"] = out["value"]
        out["sequential_roofline_frac"] = out["roofline"]["frac"]
        out["value"] = batched["pairs_per_s"]
        out["ms_per_step"] = 1e3 * batched["batch"] / batched["pairs_per_s"]
        out["config"]["workload"] += f"; value = calc_batch of {batched['batch']} pairs"
        out["roofline"]["achieved"] = algo * batched["pairs_per_s"] / 1e9
        out["roofline"]["frac"] = algo * batched["pairs_per_s"] / 1e9 / HBM_PEAK_GBS
        out["roofline"]["note"] = ("batched calc (blockIdx.z = pair): bytes = fused-iteration accounting (SURVEY 8d); pair groups of 4 on two "
                                   "streams keep a level's 22 planes per pair in the last-level cache; the level-0 iteration launch moves 1.02 x its "
                                   "compulsory bytes past the L2, at 4.5-5 TB/s (profiles/r16, r17z)")
    # four independent objects on four streams (distinct handles share nothing: the reference's constant-memory race does not exist
    # here): a 640 x 480 pair alone cannot fill 256 CUs, concurrent pairs can
    try:
        ns = 4
        streams = [torch.cuda.Stream() for _ in range(ns)]
        algs = [cuda.FarnebackOpticalFlow.create() for _ in range(ns)]
        flows = [torch.empty_like(flow) for _ in range(ns)]
        torch.cuda.synchronize()
        for rep in range(2):                       # first round warms the handles up
            tq = time.perf_counter()
            for _ in range(max(1, args.steps * n // ns)):
                for k in range(ns):
                    with torch.cuda.stream(streams[k]):
                        algs[k].calc(t0_, t1_, flows[k])
            torch.cuda.synchronize()
            el4 = time.perf_counter() - tq
        if not all(torch.equal(fk, flow) for fk in flows):
            raise RuntimeError("concurrent objects disagree with the sequential result")
        out["four_streams_pairs_per_s"] = max(1, args.steps * n // ns) * ns / el4
    except Exception as e:   # never at the expense of the line above
        out["four_streams_pairs_per_s"] = {"error": repr(e)[:200]}
    # the third dense flow class of the module on the same pair: DensePyrLKOpticalFlow((13, 13), 3, 30)
    lk = cuda.DensePyrLKOpticalFlow.create()
    lk.calc(t0_, t1_, flow)
    torch.cuda.synchronize()
    tq = time.perf_counter()
    for _ in range(10):
        lk.calc(t0_, t1_, flow)
    torch.cuda.synchronize()
    out["dense_pyrlk_pairs_per_s"] = 10 / (time.perf_counter() - tq)
    out["dense_pyrlk_epe_vs_analytic_flow_px"] = float(synth.epe(flow.cpu().numpy()[40:-40, 40:-40], gt[40:-40, 40:-40]))
    # and the sparse sibling: 10 000 points on the same pair, SparsePyrLKOpticalFlow((21, 21), 3, 30)
    try:
        rng = np.random.default_rng(0)
        pts = torch.from_numpy(np.stack([rng.uniform(0, W, 10000), rng.uniform(0, H, 10000)], 1).astype(np.float32)).to(dev)
        slk = cuda.SparsePyrLKOpticalFlow.create()
        slk.calc(t0_, t1_, pts)
        torch.cuda.synchronize()
        tq = time.perf_counter()
        for _ in range(10):
            slk.calc(t0_, t1_, pts)
        torch.cuda.synchronize()
        out["sparse_pyrlk_points_per_s"] = 10 * 10000 / (time.perf_counter() - tq)
    except Exception as e:
        out["sparse_pyrlk_points_per_s"] = {"error": repr(e)[:200]}
    if not args.no_cpu:
        from oracle import oracle as O
        tq = time.perf_counter()
        O.pyrlk_dense(I0, I1)
        lk_cpu = 1.0 / (time.perf_counter() - tq)
        O.fb_calc(I0, I1)
        t0 = time.perf_counter()
        k = 0
        while k < 3 or time.perf_counter() - t0 < 5.0:
            O.fb_calc(I0, I1)
            k += 1
        ct = (time.perf_counter() - t0) / k
        out["cpu_baseline"] = {"value": 1.0 / ct, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"{k} x 1 pair {W}x{H}, {ct * 1e3:.0f} ms each, oracle/farneback_ref.c (OpenMP rows)",
                               "dense_pyrlk_pairs_per_s": lk_cpu}
    return out


def bench_surf(args):
    """BASELINE configs[3]: SURF detect+describe on a 3840x2160 frame, hessianThreshold=400 (defaults otherwise)."""
    import numpy as np
    import torch
    from opencv_contrib_amd import cuda, synth
    dev = torch.device("cuda", 0)
    W, H = args.width, args.height
    img = synth.blob_image(H, W, seed=7)
    t = torch.from_numpy(img).to(dev)
    alg = cuda.SURF_CUDA.create(400.0)
    n = max(args.batch, 1)
    for _ in range(args.warmup * n):
        kp, desc = alg.detectWithDescriptors(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps * n):
        kp = alg.detect(t)
    torch.cuda.synchronize()
    el_det = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.steps * n):
        kp, desc = alg.detectWithDescriptors(t)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nf = int(kp.shape[1])
    # detector bytes: integral 1 B in + 4 B out per px; det+trace 8 B x (layers+2) per sample, NMS reads 4 B x 3 layers
    algo = 0.0
    for o in range(4):
        px = (W >> o) * (H >> o)
        algo += px * (8 * 4 + 12 * 2)
    algo += W * H * 5.0
    out = {"metric": "frames/sec SURF detect+describe @4K", "value": args.steps * n / el, "unit": "frames/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/f64/f32", "data": "synthetic",
           "config": {"workload": f"SURF_CUDA(400, 4 octaves, 2 layers, 64-d, oriented) on {W}x{H} CV_8UC1 blob image "
                                  f"(BASELINE configs[3]), {n} frames/step", "features": nf},
           "detect_only_frames_per_s": args.steps * n / el_det, "features_per_s": nf * args.steps * n / el,
           # The detector is NOT under the HBM roofline (its tables are L2 / MALL resident): the HBM figures are kept as a view, the bound
           # named is what the stage is made of -- dependent gathers from the integral table (octaves 1-3: 32 shared-corner taps per
           # sample; octave 0 from LDS tiles) and the f64 box arithmetic
           "roofline": {"bound": "gather_latency_and_f64_valu (not hbm)", "achieved": algo * args.steps * n / el_det / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s (HBM view)",
                        "frac": algo * args.steps * n / el_det / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("surf_det_trace")[0],
                        "traffic_kernel": "k_det_trace (one octave per launch: the stage-level form; the default path runs all octaves in k_det_trace_all), bytes per launch, mean over the octaves",
                        "traffic_source": pmc_traffic("surf_det_trace")[1],
                        "note": "detector stage only; round 3: 32 shared-corner taps per sample with wave-uniform offsets, box sums in u32, "
                                "division by the box area as reciprocal + one fma correction (exact), XCD-contiguous band order; the row "
                                "scan of the non-maximum suppression split over 8 waves per 4K row (profiles/r03k: k_det_trace 159 -> 99 us, "
                                "k_nms_flag 91 -> 54 us per octave launch).  Not HBM-bound (SURVEY 8d config 4): the integral table "
                                "(33 MB) is L2 / MALL resident; what is left is the f64 arithmetic of the box sums and the dependent gathers; the fused "
                                "all-octave path needs six launches per frame"}}
    # Binding roofline of the detector and the descriptor kernels, MEASURED (round 4): tag lookups of the per-CU vector
    # L1 (TCP_TOTAL_CACHE_ACCESSES, rocprofv3 --pmc, profiles/surf_counters.json) over the kernels' launch durations against one lookup
    # per clock and CU.  k_det_trace_all runs at 0.73 of that rate, k_descriptors at 0.84, the LDS-staged large-feature kernel at 0.46
    # (a quarter of its time the L1 is stalled on pending misses): the gathers, not HBM and not VALU, are what these kernels are made of.
    try:
        sc = json.load(open(os.path.join(ROOT, "profiles", "surf_counters.json")))
        kern = {k: v for k, v in sc["kernels"].items() if v.get("frac_of_l1_tag_peak") is not None and v.get("avg_us")}
        # the object's bound is the BINDING one -- L1 tag lookups of the kernel the frame spends most of its time in -- and the
        # HBM figures move to `hbm_view` (the detector's tables are cache resident: its HBM fraction describes nothing)
        bk = max(kern, key=lambda k: kern[k]["avg_us"])
        hbm_view = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_kernel", "traffic_source", "note")}
        out["roofline"] = {"bound": "l1_tag_lookups", "kernel": bk, "achieved": kern[bk]["l1_tag_lookups_per_s"] / 1e9, "peak": sc["peak_l1_tag_lookups_per_s"] / 1e9,
                           "unit": "G tag lookups/s", "frac": kern[bk]["frac_of_l1_tag_peak"], "avg_launch_us": kern[bk]["avg_us"],
                           "traffic": None, "lines_per_wave_load": kern[bk].get("lines_per_wave_load"), "source": sc["source"],
                           "kernels": {k: {"frac": v["frac_of_l1_tag_peak"], "lookups_per_launch": v["tcp_total_cache_accesses"], "avg_launch_us": v["avg_us"],
                                           "lines_per_wave_load": v.get("lines_per_wave_load"), "l1_hit_rate": v.get("l1_hit_rate")} for k, v in kern.items()},
                           "hbm_view": hbm_view}
    except Exception as e:
        out["roofline"]["l1_tag_lookups"] = {"error": repr(e)[:200]}
    # the descriptor half of a frame (k_orientation + k_descriptors / k_descriptors_staged on the keypoints just found): its bound is the
    # gather, not HBM and not VALU (r03l: halving the per-texel arithmetic changed nothing).  A patch sample of a feature of scale s reads
    # ~s x s texels of the ROTATED window.  Small features (s < 5): the 64 lanes of a wave sit in 64 different cells of the 21 x 21 patch, so
    # a wave-level byte load touches 64 different 64-B lines.  Large features (s >= 5, round 3): the texels are staged through LDS by lanes
    # arranged as 8 x 8 blocks of the window lattice; a rotated 8 x 8 block covers at most 12 image rows and straddles a line boundary in
    # a few of them: modelled as 14 lines per 64 texels.  One line per clock and CU through the vector L1 is the peak.
    try:
        kd = cuda.SURF_CUDA.downloadKeypoints(kp)
        sc = np.maximum(np.asarray(kd["size"], np.float64) * 1.2 / 9.0, 1.0)
        tex = 441.0 * (sc * sc + 2.0 * sc)      # s x s interior texels + the fractional border rows / columns of a cell
        stage_s = float(os.environ.get("MIFLOW_SURF_STAGE_S", "5") or 5)
        staged = (sc >= stage_s) if stage_s > 0 else np.zeros_like(sc, bool)
        texels = float(tex.sum())
        lines = float(tex[~staged].sum() + tex[staged].sum() * 14.0 / 64.0)
        for _ in range(2):
            alg.detectWithDescriptors(t, keypoints=kp, useProvidedKeypoints=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps * n):
            alg.detectWithDescriptors(t, keypoints=kp, useProvidedKeypoints=True)
        torch.cuda.synchronize()
        el_desc = (time.perf_counter() - t0) / (args.steps * n)
        peak_lines = 256 * 2.4e9
        out["descriptor_roofline"] = {"bound": "l1_gather_lines", "modelled": True,
                                      "measured": "roofline.l1_tag_lookups: k_descriptors 0.84, k_descriptors_staged 0.46 of the L1 tag-lookup rate; the model below "
                                                  "under-counts the staged kernel's lookups (42 measured per wave load against 14 modelled)",
                                      "kernel": "k_descriptors / k_descriptors_staged (+ k_orientation, integral): orientation and 64-d "
                                      "descriptors of the frame's keypoints (useProvidedKeypoints)",
                                      "achieved": lines / el_desc / 1e9, "peak": peak_lines / 1e9, "unit": "G lines/s",
                                      "frac": lines / el_desc / peak_lines, "texel_reads_per_frame": texels, "modelled_lines_per_frame": lines,
                                      "features_staged_through_lds": int(staged.sum()), "ms_per_frame": 1e3 * el_desc,
                                      "note": "achieved = modelled 64-B line fetches per second: rotated-window texel reads (441 cells x (s^2 + 2 s) per "
                                              "feature, s = size x 1.2 / 9), one line per read for the features below the staging threshold, 14 lines "
                                              "per 64 reads for the staged ones; peak = one line per clock and CU (64 B/clk vector L1, "
                                              "MI355X_MICROARCH.md).  Before the staging (r06w) every read was its own line: 0.45 of the peak"}
    except Exception as e:
        out["descriptor_roofline"] = {"error": repr(e)[:200]}
    # the secondary's `roofline` used to be the counter view -- fractions of the L1 tag-lookup rate read from a
    # static file of an earlier counter session (0.84: saturated lookups, not efficiency; nothing of it measured in this run).  The
    # object's bound is now the descriptor half's MODELLED line fetches over the time measured HERE (a frame spends most of its time
    # there); the counter session stays beside it as `counter_view`, labelled as what it is.
    try:
        dr = out["descriptor_roofline"]
        if "error" not in dr:
            cv = out["roofline"]
            if isinstance(cv, dict):
                cv["not_measured_in_this_run"] = "profiles/surf_counters.json: per-launch means of the rocprofv3 --pmc / --kernel-trace session named in `source`"
            out["roofline"] = {"bound": "l1_gather_lines (modelled lines, time measured in this run)", "kernel": "k_orientation + k_descriptors_staged (descriptor half of a frame, provided keypoints)",
                               "achieved": dr["achieved"], "peak": dr["peak"], "unit": dr["unit"], "frac": dr["frac"], "traffic": None,
                               "avg_launch_us": 1e3 * dr["ms_per_frame"], "measured_in_run": True,
                               "frame_ms": {"detect": 1e3 * el_det / (args.steps * n), "detect_describe": 1e3 * el / (args.steps * n), "describe_given_keypoints": dr["ms_per_frame"]},
                               "counter_view": cv}
    except Exception as e:
        out["roofline_restructure_error"] = repr(e)[:200]
    # two handles on two streams, frames alternating: distinct handles share nothing (the reference serialises every SURF_CUDA call
    # process-wide with a static mutex, surf.cuda.cpp:117,371,383), so one frame's descriptor kernel overlaps the next frame's detector
    try:
        streams = [torch.cuda.Stream() for _ in range(2)]
        algs = [cuda.SURF_CUDA.create(400.0) for _ in range(2)]
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                algs[i].detectWithDescriptors(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps * n * 2):
            with torch.cuda.stream(streams[i & 1]):
                kp2, desc2 = algs[i & 1].detectWithDescriptors(t)
        torch.cuda.synchronize()
        out["two_handles_two_streams_frames_per_s"] = args.steps * n * 2 / (time.perf_counter() - t0)
        out["two_handles_same_result"] = bool(torch.equal(desc2, desc))
        del algs, streams
    except Exception as e:
        out["two_handles_two_streams_frames_per_s"] = {"error": repr(e)[:200]}
    # the step after detect/describe: brute-force 2-NN matching of the frame's descriptors against themselves
    bfm = cuda.createBFMatcher()
    bfm.knnMatchDevice(desc, desc, k=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        bfm.knnMatchDevice(desc, desc, k=2)
    torch.cuda.synchronize()
    elm = (time.perf_counter() - t0) / 3
    out["bf_knn2_match_ms"] = 1e3 * elm
    out["bf_descriptor_pairs_per_s"] = float(desc.shape[0]) ** 2 / elm
    # the matcher is VALU work, not HBM work: one subtract and one fma per descriptor element and pair, against the f32 issue peak
    # of 256 CUs x 64 lanes x 2.4 GHz (MI355X_MICROARCH.md)
    valu_peak = 256 * 64 * 2.4e9
    out["bf_roofline"] = {"bound": "valu", "achieved": float(desc.shape[0]) ** 2 * desc.shape[1] * 2 / elm / 1e12, "peak": valu_peak / 1e12,
                          "unit": "T lane-op/s", "frac": float(desc.shape[0]) ** 2 * desc.shape[1] * 2 / elm / valu_peak}
    if not args.no_cpu:
        from oracle import oracle as O
        t0 = time.perf_counter()
        r = O.surf_detect_describe(img, O.surf_params(hessian_threshold=400.0))
        ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / ct, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"1 frame {W}x{H}, {r['n']} features, {ct:.1f} s wall, oracle/surf_ref.c (OpenMP in det/trace and descriptors)"}
        # round 4: the reference's OWN CPU class (xfeatures2d/src/surf.cpp compiled verbatim, oracle/_ref/libref_surfcpu.so; the stub's
        # parallel_for_ runs its invokers on OpenMP stripes) on the same frame with the same octaves / layers -- the baseline of kind
        # "reference"; the port above (the CUDA class's arithmetic on the host) stays beside it
        try:
            from oracle import refocl
            if os.path.exists(refocl.surfcpu_lib_path()):
                t0 = time.perf_counter()
                kp_c, _ = refocl.surfcpu_detect_and_compute(img, 400.0, 4, 2, extended=False, upright=False)
                cr = time.perf_counter() - t0
                port = out["cpu_baseline"]
                out["cpu_baseline"] = {"value": 1.0 / cr, "unit": "frames/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": f"1 frame {W}x{H}, {len(kp_c)} features, {cr:.1f} s wall, cv::xfeatures2d::SURF (surf.cpp verbatim, stub core, "
                                                 "OpenMP stripes), hessianThreshold 400, 4 octaves x 2 layers",
                                       "port_of_the_cuda_class": port}
        except Exception as e:
            out["cpu_baseline"]["reference_cpu_class_error"] = repr(e)[:200]
    return out


COMPACT_LIMIT = 4096   # bytes of the LAST stdout line (the driver's record keeps a bounded tail; r04's 21.9 KB line did not parse)


def _finite(x):
    """Strict JSON: non-finite floats become null; floats keep 6 significant digits in the compact line."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.6g" % x)
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def _s(x, n=110):
    x = str(x)
    return x if len(x) <= n else x[:n - 1] + "~"


def _pick(d, *keys):
    d = d or {}
    return {k: d[k] for k in keys if k in d and d[k] is not None}


def miflow_env():
    """Every MIFLOW_* variable of the process environment (the library reads its switches from there): part of the record."""
    return {k: _s(v, 40) for k, v in sorted(os.environ.items()) if k.startswith("MIFLOW_")}


def compact_line(out, full_path=None):
    """The one line the driver parses: at most COMPACT_LIMIT bytes of strict JSON with the contract's keys, the roofline and
    cpu_baseline objects, the accuracy figures and the environment.  The long form (variants, secondary workloads, per-level
    tables, notes) goes to `full_path` and to stderr.  Optional groups are dropped, last first, if the line would not fit."""
    r = out.get("roofline") or {}
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config") or {}
    c["config"] = _pick(cfg, "iterations", "epsilon", "warps", "nscales", "executed_iterations_per_warp_mean", "semantics", "math", "lanes",
                        "algorithmic_GB_per_pair", "batch", "ndisp", "block_size", "width", "height")
    c["config"]["workload"] = _s(cfg.get("workload", ""), 118)
    if r:
        roof = _pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_timed")
        roof["kernel"] = _s(r.get("kernel", ""), 60)
        h = r.get("hbm") or {}
        roof.update({("hbm_" + k): h[k] for k in ("algorithmic_bytes_per_launch_mean", "traffic_GBps", "traffic_frac_of_hbm_peak") if h.get(k) is not None})
        if isinstance(r.get("rate_weighted"), dict) and "frac_at_2.4GHz" in r["rate_weighted"]:
            roof["rate_weighted_frac"] = r["rate_weighted"]["frac_at_2.4GHz"]
        if isinstance(r.get("isolated_one_lane"), dict):
            roof["alone_frac"] = r["isolated_one_lane"].get("frac")
        c["roofline"] = roof
    if out.get("cpu_baseline"):
        cb = out["cpu_baseline"]
        c["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind", "physical_cores", "cpu_quota_cores")
        c["cpu_baseline"]["sample"] = _s(cb.get("sample", ""), 118)
    for k in ("epe_vs_cpu_ref_px", "ccorr_dissimilarity_vs_cpu_ref", "epe_vs_analytic_flow_px", "gathered_flows_identical"):
        if out.get(k) is not None:
            c[k] = out[k]
    rr = out.get("rccl_ranks") or {}
    if rr:
        c["rccl_ranks"] = _pick(rr, "world_size", "distinct_devices")
        c["rccl_ranks"]["backend"] = _s(rr.get("backend"), 40) if rr.get("backend") else None
    if out.get("per_rank_pairs_per_s"):
        c["per_rank_pairs_per_s"] = out["per_rank_pairs_per_s"]
    c["env"] = miflow_env()
    c["full"] = full_path
    # optional groups, most important first; dropped from the end if the line would exceed the limit
    opt = []
    if isinstance(r.get("second_kernel"), dict):
        sk = r["second_kernel"]
        d = _pick(sk, "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "traffic")
        d["kernel"] = _s(sk.get("kernel", ""), 40)
        roof["second_kernel"] = d   # part of the roofline object (bounded: ~250 bytes), not an optional group
    if isinstance(out.get("whole_job_hbm_traffic"), dict):
        opt.append(("whole_job_hbm_traffic", _pick(out["whole_job_hbm_traffic"], "GB_per_pair", "GBps", "frac_of_hbm_peak")))
    if isinstance(out.get("power"), dict):
        opt.append(("power", _pick(out["power"], "sclk_MHz", "socket_power_W", "roofline_frac_at_measured_clock")))
    if isinstance(out.get("with_scatter_gather"), dict):
        opt.append(("with_scatter_gather", {k: (_s(v, 80) if isinstance(v, str) else v) for k, v in out["with_scatter_gather"].items()
                                            if not isinstance(v, (dict, list))}))
    sec = out.get("secondary") or {}
    if sec:
        d = {}
        for name, e in sec.items():
            if not isinstance(e, dict):
                continue
            if "error" in e:
                d[name] = {"error": _s(e["error"], 60)}
                continue
            x = _pick(e, "value", "unit", "sequential_compute_pairs_per_s", "sequential_calc_pairs_per_s", "detect_only_frames_per_s")
            rf = e.get("roofline") or {}
            x.update({("roofline_" + k): (_s(rf[k], 30) if isinstance(rf[k], str) else rf[k]) for k in ("bound", "frac") if rf.get(k) is not None})
            if isinstance(e.get("cpu_baseline"), dict):
                x["cpu"] = e["cpu_baseline"].get("value")
            d[name] = x
        opt.append(("secondary", d))
    var = out.get("variants") or {}
    if var:
        d = {}
        for name, e in var.items():
            if isinstance(e, dict):
                v = e.get("pairs_per_s", e.get("calcs_per_s"))
                if v is not None:
                    d[_s(name, 50)] = v
                for k2 in ("calcs_per_s_three_scenes_in_turn",):
                    if e.get(k2) is not None:
                        d[_s(name, 48) + ":3scenes"] = e[k2]
        opt.append(("variants_pairs_per_s", d))
    for k, v in opt:
        c[k] = v
    c = _finite(c)
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    while len(line.encode()) > COMPACT_LIMIT and opt:
        k, _ = opt.pop()
        c.pop(k, None)
        c["dropped"] = c.get("dropped", []) + [k]
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line.encode()) > COMPACT_LIMIT:   # cannot happen with the bounded strings above; never print an over-long line
        c = {k: c[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data", "roofline", "cpu_baseline", "full") if k in c}
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    return line


def emit(out, name="bench_full.json"):
    """Long form -> bench_full.json (+ gpurun_out/ when that scratch directory exists) and stderr; compact form = the ONLY stdout line."""
    full_path = None
    txt = json.dumps(_finite_keep(out))
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, name), "w") as f:
                    f.write(txt + "\n")
                full_path = full_path or os.path.relpath(os.path.join(d, name), ROOT)
        except OSError:
            pass
    sys.stderr.write("bench.py full record: " + txt + "\n")
    sys.stderr.flush()
    print(compact_line(out, full_path), flush=True)


def _finite_keep(x):
    """The long form with full precision; only NaN / inf replaced (strict JSON)."""
    if isinstance(x, float):
        return None if (x != x or x in (float("inf"), float("-inf"))) else x
    if isinstance(x, dict):
        return {str(k): _finite_keep(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite_keep(v) for v in x]
    return x


def static_mix_rate_weighted():
    """SIMD cycles of one pipeline stage (one pixel row of one wave) of the T = 10 kernel at the measured per-operation issue rates."""
    try:
        jw = tbr_jw()
        return float(json.load(open(os.path.join(ROOT, "profiles", "static_mix_tbr.json" if jw >= 2 else "static_mix_tbr_jw0.json")))["rate_weighted_per_stage"]["cycles"])
    except Exception:
        return 145.84


def static_mix():
    """Per pipeline stage and pixel row of the T = 10 kernel that runs (tools/static_mix.py): profiles/static_mix_tbr.json = the default
    k_iterate_tbr<10, 1, ., 4, 2, 0, 2> (joined waves, barrier form), profiles/static_mix_tbr_jw0.json = the independent-wave form."""
    jw = tbr_jw()
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "static_mix_tbr.json" if jw >= 2 else "static_mix_tbr_jw0.json")))["per_stage_and_pixel"]
    except Exception:
        return ({"valu_plain": 35.338, "transcendental": 4.069, "dpp": 4.0, "cndmask": 1.985} if jw >= 2 else
                {"valu_plain": 31.662, "transcendental": 4.069, "dpp": 4.0, "cndmask": 2.223})


def pmc_traffic(key, pairs_per_launch=None, sub=None):
    """Measured HBM bytes per launch (separate rocprofv3 --pmc passes of this command, tools/pmc_summary.py), or None.  The TV-L1
    figures were collected at `pairs_per_launch` pairs per kernel launch (recorded in the file); they scale with the pairs a launch
    of the current run processes."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if sub:   # StereoBM: launches of one pair / of a batch kept apart (tools/pmc_secondary.py)
            e = tj[key][sub]
            return e.get("hbm_bytes_per_pair", e["hbm_bytes"]), tj[key].get("source")
        v = tj[key]["hbm_bytes_per_launch"]
        if pairs_per_launch and v and tj[key].get("pairs_per_launch"):
            v = v * pairs_per_launch / tj[key]["pairs_per_launch"]
        return v, tj[key].get("source")
    except Exception:
        return None, None


def pmc_pairs(key, default):
    """Pairs per launch the PMC figure of `key` was collected at (profiles/pmc_traffic.json)."""
    try:
        return int(json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[key].get("pairs_per_launch", default))
    except Exception:
        return default


def secondary(args):
    """The other BASELINE configs in the same driver-run line (reduced step counts): configs[0] Farneback 640x480,
    configs[2] StereoBM 1080p/128/15, configs[3] SURF 4K; each with its bound, fraction and CPU baseline."""
    import copy
    out = {}
    for name, fn, kw in (("stereobm_1080p_nd128_bs15", bench_stereobm, dict(width=1920, height=1080, batch=64, steps=2, warmup=1)),
                         ("farneback_640x480", bench_farneback, dict(width=640, height=480, batch=16, steps=3, warmup=1)),
                         ("surf_4k_thr400", bench_surf, dict(width=3840, height=2160, batch=2, steps=3, warmup=1))):
        a = copy.copy(args)
        for k, v in kw.items():
            setattr(a, k, v)
        try:
            r = fn(a)
            keep = {k: r[k] for k in r if k not in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "warmup", "steps")}
            out[name] = keep
        except Exception as e:   # never at the expense of the headline
            out[name] = {"error": repr(e)[:300]}
    return out


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def self_launch(n):
    """`python bench.py --gpus N` without an external launcher: re-exec this command line under torch.distributed.run with N
    ranks on this node (one process per GPU, rendezvous on 127.0.0.1 and a free port) and hand its exit code back."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer buffers need it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))   # torchrun would otherwise pin every rank to 1 thread
    return subprocess.call(cmd, env=env)


def rank_identity(torch, dev, dry):
    """What one rank contributes to `rccl_ranks`: enough to see that N ranks sit on N DIFFERENT GPUs."""
    import socket
    d = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(),
         "host": socket.gethostname()}
    if not dry:
        pr = torch.cuda.get_device_properties(dev)
        d.update({"device_index": dev.index, "name": pr.name,
                  "uuid": str(getattr(pr, "uuid", "")) or None,
                  "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                  "hbm_GB": round(pr.total_memory / 2 ** 30, 1)})
    return d


def gather_rank_identities(dist, world, backend, me):
    ids = [me]
    if dist is not None:
        ids = [None] * world
        dist.all_gather_object(ids, me)
    keys = [i.get("uuid") or i.get("pci") or (i["host"], i["pid"]) for i in ids]
    return {"world_size": world, "backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if dist is not None else None,
            "ranks": ids, "distinct_devices": len(set(map(str, keys)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["tvl1", "stereobm", "farneback", "surf"], default="tvl1")
    ap.add_argument("--ndisp", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=15)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step.  64 = one GPU's share of BASELINE configs[4] (512 pairs over 8 GPUs), so the N = 1 line and the N = 2 / 4 / 8 lines run the same per-GPU workload; 16 was the step of the r01 / early r02 records, 32 that of r02z .. r04z (r05e / r05f with the joined-wave kernel: 32 | 48 | 64 | 96 | 128 pairs = 1 296 | 1 263-1 277 | 1 313-1 377 | 1 299 | 1 306 pairs/s)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--epsilon", type=float, default=0.0)
    ap.add_argument("--defaults", action="store_true", help="class defaults: 300 iterations, epsilon 0.01")
    ap.add_argument("--exact-math", action="store_true",
                    help="IEEE divide + f64 hypot, one iteration per HBM pass (oracle-faithful to ~1e-6 px); the library default is "
                         "fast math: v_rcp/v_sqrt + temporal blocking, parity-tested at mean EPE <= 5e-3 px")
    ap.add_argument("--time-block", type=int, default=0, help="iterations fused per HBM pass (fast math; 0 = auto, 1 = off)")
    ap.add_argument("--lanes", type=int, default=0, help="internal streams a batch is split over (0 = library default: 2 from 4 pairs on)")
    ap.add_argument("--semantics", type=int, default=None, help="0 = CPU class arithmetic (library default), 1 = cv::cuda's kernels")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the clock / socket-power leg (2.5 s of the same step with rocm-smi polled beside it)")
    ap.add_argument("--stop-slack", type=int, default=0, help="mi_tvl1_params.stop_slack (miflow extension; 0 = the reference's exact stopping point)")
    ap.add_argument("--cpu-iterations", type=int, default=None)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: bring the N ranks up, all-gather their identities, print the JSON line "
                         "without touching a GPU (the CPU test of the --gpus N path uses it over gloo)")
    args = ap.parse_args()
    if args.defaults:
        args.iterations, args.epsilon = 300, 0.01
    # --gpus N IS the number of ranks.  Started by a launcher (WORLD_SIZE in the environment: the driver's
    # `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) the two must agree; started bare with N > 1
    # (`python bench.py --gpus 8`) this process becomes the launcher of N ranks, one per GPU, and relays their exit code.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    # a timed library that the environment can tell to skip work or to change results is not a benchmark: the release library does
    # not read these switches at all (experiments build only); refuse them here as well so no record is ever made under them
    for bad in ("MIFLOW_X_SKIP", "MIFLOW_TB_P16"):
        if os.environ.get(bad, "0") not in ("", "0"):
            sys.exit(f"bench.py: {bad} is set -- a work-skipping / result-changing experiment switch; refusing to time under it")
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        if args.gpus > 1:
            return sys.exit(self_launch(args.gpus))
    elif int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks; "
                 f"refusing to report a line whose n_gpus is not what was asked for")
    if args.workload != "tvl1" and args.gpus != 1:
        sys.exit("bench.py: the secondary workloads are single-GPU lines (--gpus 1); the sharded metric is --workload tvl1")
    if args.workload == "stereobm":
        return emit(bench_stereobm(args), "bench_full_stereobm.json")
    if args.workload == "surf":
        if (args.width, args.height) == (1920, 1080):
            args.width, args.height = 3840, 2160
        return emit(bench_surf(args), "bench_full_surf.json")
    if args.workload == "farneback":
        if (args.width, args.height) == (1920, 1080):
            args.width, args.height = 640, 480
        return emit(bench_farneback(args), "bench_full_farneback.json")

    import numpy as np
    W, H, B = args.width, args.height, args.batch
    base_pairs = None if args.dry_run else gen_base_pairs(min(B, 16), H, W)   # forked workers: before torch touches the GPU
    import torch

    # one process per GPU; every rank runs its own batch of independent pairs (no data-path collective);
    # RCCL ("nccl") only carries the barriers and the max-over-ranks timing reduction
    from opencv_contrib_amd import parallel
    if int(os.environ.get("RANK", "0")) != 0:
        os.dup2(2, 1)   # only rank 0 owns stdout (the one JSON line); whatever the other ranks' libraries print goes to stderr
    # MIFLOW_BENCH_BACKEND=gloo + MIFLOW_BENCH_DEVICE=0: development aid, several ranks of the launcher on ONE GPU (barriers and the
    # timing reduction over gloo on the CPU) -- exercises the multi-process control flow where no multi-GPU node is at hand
    backend = os.environ.get("MIFLOW_BENCH_BACKEND", "nccl")
    dist, rank, world, local = parallel.init_distributed(backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    if "MIFLOW_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MIFLOW_BENCH_DEVICE"])
    dev = torch.device("cuda", local)
    if args.dry_run:
        ranks_info = gather_rank_identities(dist, world, backend, rank_identity(torch, dev, True))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "gpus_arg": args.gpus, "rccl_ranks": ranks_info}), flush=True)
        return
    torch.cuda.set_device(dev)
    ranks_info = gather_rank_identities(dist, world, backend, rank_identity(torch, dev, False))
    if world > 1 and backend == "nccl" and ranks_info["distinct_devices"] != world:
        sys.exit(f"bench.py: {world} ranks but only {ranks_info['distinct_devices']} distinct GPUs: {ranks_info['ranks']}")
    from opencv_contrib_amd import capi, cuda, synth

    I0, I1, base = make_inputs(B, H, W, dev, base=base_pairs)
    flows = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
    warps = 5

    def create(iterations, epsilon, **kw):
        # A DEFAULT-CONSTRUCTED object with the reference test's two setter calls (setNumIterations / setEpsilon); the miflow
        # extensions stay at the library defaults (CPU-class arithmetic, fast math, automatic fusing and lanes) unless a variant
        # or a command-line flag names them
        ext = dict(exactMath=True if args.exact_math else None, timeBlock=args.time_block, lanes=args.lanes, semantics=args.semantics, stopSlack=args.stop_slack)
        ext.update(kw)
        return cuda.OpticalFlowDual_TVL1.create(iterations=iterations, epsilon=epsilon, **ext)

    def run(iterations, epsilon, steps, warmup, profile=False, inputs=None, out=None, **kw):
        alg = create(iterations, epsilon, **kw)
        alg.setProfiling(profile)
        a0, a1 = inputs if inputs is not None else (I0, I1)
        el = time_steps(alg, a0, a1, flows if out is None else out, steps, warmup, dist)
        el = parallel.max_over_ranks(dist, el, dev if backend == "nccl" else None)
        prof = (alg.getProfile(0), alg.getProfile(1)) if profile else None
        its = alg.lastIterations(0)
        return float(el), prof, its, alg

    el, prof, its, alg = run(args.iterations, args.epsilon, args.steps, args.warmup, profile=True)
    local_s = parallel.all_over_ranks(dist, getattr(time_steps, "local_s", el), dev if backend == "nccl" else None)
    P = alg._p
    pairs = B * world * args.steps
    fps = pairs / el
    # accuracy of what was just computed: EPE vs the analytic flow of pair 0
    gt = base[0][2]
    f0 = flows[0].cpu().numpy()
    epe_gt = float(np.sqrt(((f0 - gt) ** 2).sum(-1))[40:-40, 40:-40].mean())

    mean_it = float(np.mean(its))
    ab_pair = algo_bytes_per_pair(W, H, warps, mean_it)
    blocked = (not P.exact_math) and P.time_block != 1
    (ms_it, n_it, bytes_it), (ms_w, n_w, bytes_w) = prof
    px_levels = float(sum(level_pixels(W, H)))
    px_iter_timed = px_levels * B * warps * mean_it            # pixel-iterations inside the timed iteration regions of one calc
    if args.epsilon > 0:
        bytes_it *= mean_it / args.iterations
    hbm_it = {"algorithmic_bytes_per_launch_mean": bytes_it / max(n_it, 1),
              "algorithmic_GBps": bytes_it / (ms_it * 1e-3) / 1e9 if ms_it > 0 else None,
              "note": "64 B x px x iterations executed by the launch (SURVEY 8d) / launch time; with T iterations per HBM pass this "
                      "exceeds the HBM peak by construction -- the kernel is not under the HBM roofline, see `traffic`"}
    n_lanes_run = (2 if (P.lanes == 0 and B >= 4) or P.lanes == 2 else max(1, P.lanes))   # every lane runs its own launches
    per_lane_pairs = max(1, B // n_lanes_run)
    traffic, tsrc = pmc_traffic("tbr" if blocked else "v1", per_lane_pairs)
    if traffic:
        hbm_it.update({"traffic_bytes_per_launch": traffic, "traffic_GBps": traffic / (ms_it * 1e-3 / max(n_it, 1)) / 1e9,
                       "traffic_frac_of_hbm_peak": traffic / (ms_it * 1e-3 / max(n_it, 1)) / 1e9 / HBM_PEAK_GBS, "traffic_source": tsrc})
    if blocked and args.epsilon == 0:
        # VALU issue: lane-instruction slots the kernel executes per second against the chip's issue peak.  Slots per pixel-iteration
        # = static mix of the main loop (full-rate VALU + DPP + v_cndmask count 1, the quarter-rate v_rcp / v_sqrt count 4); lanes
        # executed per owned pixel = 64 / 44 (T = 10, 1 px/lane: 10 halo columns per side); rows streamed per owned row =
        # (R + 2T) / R with R the band height the planner chose (about 1.15-1.25 at 1080p x 16: reported by MIFLOW_TB_VERBOSE)
        mix = static_mix()
        slots = mix["valu_plain"] + mix["dpp"] + mix["cndmask"] + 4.0 * mix["transcendental"]
        jw = tbr_jw()
        lanes_per_px = 512.0 / 492.0 if jw == 3 else 256.0 / 236.0 if jw else 64.0 / 44.0   # joined waves: the T-column margin exists at the two outer edges of a 256-column strip only
        ach = px_iter_timed / (ms_it * 1e-3) * slots * lanes_per_px / 1e12
        roof = {"bound": "valu_issue", "kernel": ("k_iterate_tbr<10,1,.,4,2,0,%d> (10 fused estimateU+estimateDualVariables iterations per HBM pass; " % jw) +
                ("four joined waves per 256-column strip, seam values handed over through LDS)" if jw else "independent 64-column waves)"),
                "achieved": ach, "peak": VALU_PEAK_TLIPS, "unit": "T lane-instr/s", "frac": ach / VALU_PEAK_TLIPS,
                "pixel_iterations_per_s": px_iter_timed / (ms_it * 1e-3),
                # what a plain f32 VALU stream issues on this chip with its clock MEASURED (round 4, s_memtime / s_memrealtime, tools/ubench/
                # issue_mix.hip): 2.4 SIMD cycles per wave-instruction at 8 waves / SIMD and 2.33 GHz; the 3.1 "cycles" of the earlier rounds
                # were wall time x a nominal 2.4 GHz
                # (tools/ubench/valu_rates.hip, profiles/r01p/valu_rates.txt) against the 2 of the 157.3 TF figure
                "peak_measured_plain_valu": 63.2, "frac_of_measured_peak": ach / 63.2,
                "issue_slots_per_pixel_iteration": slots, "lanes_executed_per_owned_pixel": lanes_per_px,
                "band_halo_rows_not_counted": True,
                "avg_launch_us": 1e3 * ms_it / max(n_it, 1), "launches_timed": n_it,
                "iterations_per_launch_mean": mean_it * warps * len(its) * n_lanes_run / max(n_it, 1),
                "traffic": traffic, "hbm": hbm_it,
                # against the MEASURED issue rates (round 4, profiles/valu_rates_gfx950.json): only fma / mul / add / sub and a few integer
                # operations issue in 2.4 cycles, DPP / selects / med3 take 4.2, v_rcp / v_sqrt 8.2: a stage costs 145.8 SIMD cycles, not the
                # 115 of the spec-rate slot count.  wave_stages_per_s = executed (halo lanes included) rows x stages of 64 lanes.
                "rate_weighted": {"cycles_per_wave_stage": static_mix_rate_weighted(),
                                  "peak_wave_stages_per_s_at_2.4GHz": 1024 * 2.4e9 / static_mix_rate_weighted(),
                                  "achieved_wave_stages_per_s": px_iter_timed / (ms_it * 1e-3) * lanes_per_px / 64.0,
                                  "frac_at_2.4GHz": px_iter_timed / (ms_it * 1e-3) * lanes_per_px / 64.0 / (1024 * 2.4e9 / static_mix_rate_weighted()),
                                  "note": "the same achieved rate against the stage's cycles at measured rates; the power leg's clock (power.sclk_MHz) scales the peak further"},
                "note": "HIP events on the launch streams around each warp's iteration launch; with lanes = 2 the other half batch's "
                        "kernels share the GPU during these intervals (the rocprofv3 kernel trace shows the same durations)"}
    else:
        if blocked:
            kname = "k_iterate_tbr MODE 1 (speculative steps of the convergence-checked path: block, settle, replay)"
        elif P.exact_math and args.epsilon == 0 and P.time_block != 1:
            kname = "k_iterate_tbr MODE 2 (exact math, blocks of up to 5 fused iterations, bit-identical to one launch per iteration)"
        else:
            kname = "k_iterate (fused estimateU+estimateDualVariables, one iteration per launch)"
        roof = {"bound": "hbm", "kernel": kname, "achieved": hbm_it["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (hbm_it["algorithmic_GBps"] or 0) / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_us": 1e3 * ms_it / max(n_it, 1), "launches_timed": n_it, "hbm": hbm_it}
    # second kernel: the fused-gradient warp.  Bound: HBM -- alone it moves its 44 B/px at 0.5-0.6 of the peak (r02z3, one lane; the
    # LDS-staged variant that takes the 128 B/px of gathered windows off the vector-memory path is only 11 % faster, so that path
    # is not what limits it); with the other lane's iteration kernel on the chip, as in the intervals timed here, about half that.
    # The vector-memory-path view (128 B gathered per pixel as 4-byte-aligned dwordx4 / dwordx2 + 12 B in + 16 B out against
    # 64 B/clk per CU) is kept next to it.
    warp_bytes_px = 128.0 + 12.0 + 16.0
    hbm_w = bytes_w / (ms_w * 1e-3) / 1e9 if ms_w > 0 else None
    roof_warp = {"bound": "hbm", "kernel": "k_warp6 (bicubic warp, centred gradient of I1 formed from a 6x6 window)",
                 "achieved": hbm_w, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_w / HBM_PEAK_GBS if hbm_w else None,
                 "avg_launch_us": 1e3 * ms_w / max(n_w, 1), "launches_timed": n_w,
                 "hbm_algorithmic_GBps": hbm_w, "hbm_algorithmic_frac": hbm_w / HBM_PEAK_GBS if hbm_w else None,
                 "vector_memory_path_GBps": px_levels * B * warps * warp_bytes_px / (ms_w * 1e-3) / 1e9 if ms_w > 0 else None,
                 "vector_memory_path_peak_GBps": VMEM_PATH_PEAK_GBS,
                 "alone_one_lane": "0.5-0.6 of the HBM peak (profiles/r02z/README.md, r02z3: 173 us average launch of 16 pairs)"}
    wtraffic, wsrc = pmc_traffic("warp6", per_lane_pairs)
    if wtraffic:
        roof_warp.update({"traffic": wtraffic, "traffic_source": wsrc})
    roof["second_kernel"] = roof_warp
    per_level = []
    try:
        lp = level_pixels(W, H)
        for lv in range(len(its)):
            li_, lw_ = alg.getProfile(0, lv), alg.getProfile(1, lv)
            per_level.append({"level": lv, "pixels": lp[lv] if lv < len(lp) else None, "iterate_ms": li_[0], "iterate_launches": li_[1],
                              "warp_ms": lw_[0], "warp_launches": lw_[1],
                              "ns_per_pixel_iteration": 1e6 * li_[0] / (lp[lv] * B * warps * mean_it) if lv < len(lp) and mean_it > 0 else None})
    except Exception as e:
        per_level = [{"error": repr(e)[:200]}]
    roof["time_share"] = {"iterate_ms_per_calc": ms_it, "warp_ms_per_calc": ms_w, "calc_ms": 1e3 * el / args.steps,
                          "per_level": per_level,
                          "note": "event intervals of the two lanes add up; they overlap in wall time.  per_level: level 0 = finest; "
                                  "ns_per_pixel_iteration makes a regression at a coarse level visible"}
    if blocked and args.epsilon == 0:
        # Whole job in issue terms: lane-instruction slots of the two dominant kernels per second of WALL time (the per-kernel
        # figures above are event intervals during which the other lane's kernels share the GPU).  executed = what the SIMDs run:
        # owned pixels x column halo (64/44) x band halo ((R + 2T)/R, R from the planner) for the iteration kernel + the warp's
        # static count; useful = the same without the two halo factors.
        lanes_n = out_lanes = (2 if (P.lanes == 0 and B >= 4) or P.lanes == 2 else max(1, P.lanes))
        dims = [(max(int(round(W * 0.8 ** s)), 1), max(int(round(H * 0.8 ** s)), 1)) for s in range(5)]
        per_lane = max(1, B // lanes_n)

        def band_rows(w_, h_):   # the library's own plan (mi_tvl1_query_plan); tbr_band_rows mirrors it for builds without the entry
            try:
                import ctypes as C_
                k_, r_ = C_.c_int(0), C_.c_int(0)
                capi.check(capi.lib().mi_tvl1_query_plan(w_, h_, per_lane, 10, C_.byref(k_), C_.byref(r_)))
                return float(r_.value)
            except Exception:
                return float(tbr_band_rows(w_, h_, per_lane))
        it_exec = sum(w_ * h_ * (band_rows(w_, h_) + 20.0) / band_rows(w_, h_) for w_, h_ in dims) \
            * B * warps * mean_it * slots * lanes_per_px
        it_useful = px_levels * B * warps * mean_it * slots
        wp = px_levels * B * warps * WARP_VALU_PER_PX
        step_s = el / args.steps
        roof["whole_job_valu"] = {
            "executed_T_lane_instr_per_s": (it_exec + wp) / step_s / 1e12, "useful_T_lane_instr_per_s": (it_useful + wp) / step_s / 1e12,
            "peak": VALU_PEAK_TLIPS, "frac_executed": (it_exec + wp) / step_s / 1e12 / VALU_PEAK_TLIPS,
            "frac_useful": (it_useful + wp) / step_s / 1e12 / VALU_PEAK_TLIPS,
            "frac_executed_of_measured_plain_valu_peak": (it_exec + wp) / step_s / 1e12 / 63.2,   # 63.2 T lane-instr/s: a pure v_fma stream, clock measured
            "band_rows_finest_level": band_rows(W, H), "warp_valu_per_pixel": WARP_VALU_PER_PX,
            "note": "iteration kernel (static mix x pixel-iterations x halo factors) + warp kernel (static count x pixels) over the wall "
                    "time of a step; resize / convert / pack (4 % of the kernel time) not counted"}

    out = {"metric": "frame-pairs/sec dense TV-L1 flow @1080p", "value": fps, "unit": "pairs/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"DualTVL1 dense flow, {W}x{H} CV_32FC1, {B} pairs/GPU/step (BASELINE configs[1]; 64 pairs/GPU = the per-GPU share of configs[4])",
                      "object": "default-constructed OpticalFlowDual_TVL1 + setNumIterations(N) + setEpsilon(eps); miflow extensions at "
                                "the library defaults unless listed.  NOTE: the reference's accuracy test (cudaoptflow/test/"
                                "test_optflow.cpp:448-451) calls setNumIterations(10) ONLY, i.e. epsilon stays 0.01 and its loop is "
                                "convergence-checked: that literal setting is variants.iterations10_eps0.01_reference_test_setting; the "
                                "headline's epsilon = 0 makes N = 10 fixed work (BASELINE.md section 2 accounting)",
                      "iterations": args.iterations, "epsilon": args.epsilon, "warps": warps, "nscales": 5,
                      "executed_iterations_per_warp_mean": mean_it,
                      "semantics": "CPU_REF" if P.semantics == capi.MI_SEM_CPU_REF else "CUDA_COMPAT",
                      "math": "exact" if P.exact_math else "fast", "time_block": P.time_block,
                      "lanes": alg.calc_lanes() if hasattr(alg, "calc_lanes") else (2 if (P.lanes == 0 and B >= 4) or P.lanes == 2 else 1),
                      "algorithmic_GB_per_pair": ab_pair / 1e9},
           # every rank's own rate (its batch over its own time, before the closing barrier): a slow GPU or link shows here, the
           # aggregate `value` is all pairs over the SLOWEST rank's time
           "per_rank_pairs_per_s": [B * args.steps / t_ for t_ in local_s],
           "rccl_ranks": ranks_info,
           "whole_job_algorithmic_GBps": ab_pair * fps / 1e9,
           "whole_job_frac_of_hbm_peak": ab_pair * fps / 1e9 / HBM_PEAK_GBS,
           "epe_vs_analytic_flow_px": epe_gt,
           "roofline": roof}

    if world == 1 and not args.no_power:
        # the same step held on the chip for a few seconds with rocm-smi polled beside it (untimed leg: the headline above is not touched)
        pw = power_leg(lambda: alg.calc_batch(I0, I1, flows), float(B))
        if pw and "sclk_MHz" in pw:
            f_ghz = pw["sclk_MHz"] / 1e3
            pw["note"] = ("shader clock and socket power while the step of `value` runs back to back (rocm-smi, settled part of the leg).  "
                          "MI355X holds this workload at its power limit: the clock, not the kernels' issue or HBM efficiency, is what gives "
                          "(DESIGN 4.1 round 4: a copy at 4.8 TB/s alone draws ~975 W, a pure v_fma stream at full rate ~980 W, idle ~270 W; "
                          "removing 16 % of the iteration kernel's instructions changed nothing).  Peaks at the measured clock are given "
                          "next to the 2.4 GHz spec peaks the fractions above use")
            pw["valu_peak_at_measured_clock_T_lane_instr_per_s"] = VALU_PEAK_TLIPS * f_ghz / 2.4
            if isinstance(roof.get("frac"), float) and roof.get("bound") == "valu_issue":
                pw["roofline_frac_at_measured_clock"] = roof["achieved"] / (VALU_PEAK_TLIPS * f_ghz / 2.4)
                if "rate_weighted" in roof:
                    pw["rate_weighted_frac_at_measured_clock"] = roof["rate_weighted"]["frac_at_2.4GHz"] * 2.4 / f_ghz
            pw["joules_per_pair"] = pw["socket_power_W"] / pw["rate_during_leg"]
        out["power"] = pw
    # whole-job HBM traffic: measured bytes per launch of the two dominant kernels (separate --pmc passes, profiles/pmc_traffic.json) x the
    # launches of a step, over the wall time of a step -- the quantity north_star's "fraction of HBM roofline" bar is about
    try:
        if traffic and wtraffic and n_it and n_w:
            tot = traffic * n_it + wtraffic * n_w
            out["whole_job_hbm_traffic"] = {"bytes_per_step": tot, "GBps": tot / (el / args.steps) / 1e9, "frac_of_hbm_peak": tot / (el / args.steps) / 1e9 / HBM_PEAK_GBS,
                                            "frac_of_measured_copy_rate": tot / (el / args.steps) / 1e9 / 6290.0,
                                            "GB_per_pair": tot / B / 1e9,
                                            "note": "iteration + warp launches of one step (both lanes); resize / convert / pack (~4 % of the kernel time) not counted"}
    except Exception as e:
        out["whole_job_hbm_traffic"] = {"error": repr(e)[:200]}
    exchange_hung = False
    if world == 1 and os.environ.get("MIFLOW_BENCH_EXCHANGE_FORCE"):
        # development aid: the exchange leg on ONE GPU through a single-rank RCCL group (scatter / gather to self) -- exercises the
        # group set-up, the pipeline, the watchdog and the verification where no multi-GPU node is at hand
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        dist = tdist
    if (world > 1 or os.environ.get("MIFLOW_BENCH_EXCHANGE_FORCE")) and os.environ.get("MIFLOW_BENCH_EXCHANGE", "1") != "0" and backend == "nccl":
        # The headline number above does not depend on this leg, and must not be lost to it: the leg runs under a watchdog on every
        # rank (same limit everywhere); if the point-to-point exchange does not finish, the line is printed without it and the
        # processes leave without the collective teardown.
        import threading
        box = {}

        def leg():
            try:
                torch.cuda.set_device(dev)
                box["res"] = bench_exchange(args, parallel, dist, rank, world, dev, I0, I1, flows)
            except Exception as e:
                box["res"] = {"error": repr(e)[:300]}

        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(float(os.environ.get("MIFLOW_BENCH_EXCHANGE_TIMEOUT", "180")))
        exchange_hung = th.is_alive()
        out["with_scatter_gather"] = {"error": "timed out (watchdog); skipped"} if exchange_hung else box.get("res")
        out["gathered_flows_identical"] = (out["with_scatter_gather"] or {}).get("gathered_flows_identical")

    # the object of the timed run is released here: every variant below creates its own (a handle owns an internal stream, and
    # streams of live handles share the few hardware queues of the device)
    del alg
    if not args.no_variants and world == 1:
        var = {}
        hs = max(1, args.steps // 2)

        def vrun(name, it, eps, inputs=None, warm=1, **kw):
            try:
                e2, _, its2, _ = run(it, eps, hs, warm, inputs=inputs, **kw)
                m2 = float(np.mean(its2))
                ab = algo_bytes_per_pair(W, H, warps, m2)
                var[name] = {"pairs_per_s": B * hs / e2, "executed_iterations_per_warp_mean": m2, "algorithmic_GB_per_pair": ab / 1e9,
                             "frac_of_hbm_peak": ab * (B * hs / e2) / 1e9 / HBM_PEAK_GBS}
            except Exception as e:   # never at the expense of the headline line
                var[name] = {"error": repr(e)[:200]}

        # the reference accuracy test's LITERAL setting (test_optflow.cpp:448-451): create(); setNumIterations(10) -- epsilon 0.01
        vrun("iterations10_eps0.01_reference_test_setting", 10, 0.01)
        vrun("iterations2_eps0", 2, 0.0)
        vrun("iterations30_eps0", 30, 0.0)
        # Speculative blocks, device-decided stop.  Round 4: a warp's first block is as long as the same warp of the same pair slot
        # needed in the handle's PREVIOUS calc (SpecK::h_in).  The plain variant repeats one batch, so its history is exact -- the
        # best case of a video; `_history_from_other_pairs` alternates the batch with itself rolled by one and by two pairs, so
        # every estimate comes from a different pair; `_no_history` (MIFLOW_TB_HIST=0, a subprocess: the switch is read once) is
        # the first calc of a handle / the round-3 estimates.  Flows and iteration counts are identical in all three.
        vrun("class_defaults_300_eps0.01", 300, 0.01, warm=2)
        try:
            import torch as _t
            seqs = ([I0, _t.roll(I0, 1, 0).contiguous(), _t.roll(I0, 2, 0).contiguous()], [I1, _t.roll(I1, 1, 0).contiguous(), _t.roll(I1, 2, 0).contiguous()])
            vrun("class_defaults_300_eps0.01_history_from_other_pairs", 300, 0.01, inputs=seqs, warm=2)
            del seqs
        except Exception as e:
            var["class_defaults_300_eps0.01_history_from_other_pairs"] = {"error": repr(e)[:200]}
        try:
            import subprocess
            r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-variants", "--no-cpu", "--no-secondary", "--no-power", "--iterations", "300",
                                 "--epsilon", "0.01", "--steps", str(max(2, hs)), "--warmup", "1", "--batch", str(B)], capture_output=True, text=True,
                                env=dict(os.environ, MIFLOW_TB_HIST="0"), timeout=300)
            d_ = json.loads([l for l in r_.stdout.splitlines() if l.startswith("{")][-1])
            var["class_defaults_300_eps0.01_no_history"] = {"pairs_per_s": d_["value"]}
        except Exception as e:
            var["class_defaults_300_eps0.01_no_history"] = {"error": repr(e)[:200]}
        vrun("class_defaults_300_eps0.01_stop_slack1", 300, 0.01, stopSlack=1)   # miflow extension: up to one iteration past the reference's stop
        # the other half of the reference's own test matrix: Gamma(1.0) (test_optflow.cpp:451,530-532) -- since round 6 on the blocked
        # kernel with the illumination channel (k_iterate_tbr GAM: u3, p31, p32 ride through the same pipeline); the algorithmic
        # bytes of an iteration are 96 B/px there (three u, six p) against the 64 the two-channel figure below is computed with
        vrun("iterations10_eps0_gamma1", 10, 0.0, gamma=1.0)
        vrun("class_defaults_300_eps0.01_gamma1", 300, 0.01, gamma=1.0, warm=2)
        vrun("iterations10_eps0_exact_math", 10, 0.0, exactMath=True)
        vrun("iterations10_eps0_cuda_compat_semantics", 10, 0.0, semantics=1)
        # one lane: the whole batch on the caller's stream -- the two dominant kernels timed WITHOUT the other half batch's kernels
        # sharing the GPU (what `roofline` above cannot separate)
        try:
            e1, (p_it, p_w), _, _ = run(args.iterations, args.epsilon, hs, 1, profile=True, lanes=1)
            var["iterations10_eps0_one_lane"] = {"pairs_per_s": B * hs / e1, "iterate_avg_launch_us": 1e3 * p_it[0] / max(p_it[1], 1),
                                                 "warp_avg_launch_us": 1e3 * p_w[0] / max(p_w[1], 1),
                                                 # the warp kernel with the GPU to itself: algorithmic 44 B/px over its own launch time
                                                 "warp_hbm_algorithmic_GBps": p_w[2] / (p_w[0] * 1e-3) / 1e9 if p_w[0] > 0 else None,
                                                 "warp_hbm_algorithmic_frac": p_w[2] / (p_w[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if p_w[0] > 0 else None}
            if blocked and args.epsilon == 0 and p_it[0] > 0:
                ach1 = px_iter_timed / (p_it[0] * 1e-3) * slots * lanes_per_px / 1e12
                var["iterations10_eps0_one_lane"]["iterate_valu_issue_frac"] = ach1 / VALU_PEAK_TLIPS
                var["iterations10_eps0_one_lane"]["iterate_pixel_iterations_per_s"] = px_iter_timed / (p_it[0] * 1e-3)
            if "iterate_valu_issue_frac" in var["iterations10_eps0_one_lane"]:
                # the same kernel with the GPU to itself (no second lane): the figure that describes the kernel rather than the overlap
                out["roofline"]["isolated_one_lane"] = {
                    "frac": var["iterations10_eps0_one_lane"]["iterate_valu_issue_frac"],
                    "avg_launch_us": var["iterations10_eps0_one_lane"]["iterate_avg_launch_us"],
                    "pixel_iterations_per_s": var["iterations10_eps0_one_lane"]["iterate_pixel_iterations_per_s"],
                    "rate_weighted_frac_at_2.4GHz": var["iterations10_eps0_one_lane"]["iterate_pixel_iterations_per_s"] * lanes_per_px / 64.0 /
                                                    (1024 * 2.4e9 / static_mix_rate_weighted()),
                    "sq_counters": "profiles/r02p/pmc_sq_summary.md (VALU active 0.46, issue-stalled 0.30, parked on waitcnt 0.07 of the wave cycles)",
                    "note": "one lane draws ~1 250 W at ~2.26 GHz (profiles/r08/poll_bench_one_lane.txt): divide the fractions by 0.94 for the measured clock"}
        except Exception as e:
            var["iterations10_eps0_one_lane"] = {"error": repr(e)[:200]}
        # other batch sizes of the same call (16 / 32 = the steps of the earlier records; 64 = the per-GPU share of BASELINE configs[4])
        if B == 64:   # the headline IS that figure
            out["configs4_per_gpu_share_64_pairs"] = {"pairs_per_s": out["value"] / max(1, out["n_gpus"]), "pairs_per_step": 64, "steps": out["steps"],
                                                      "ms_per_step": out["ms_per_step"], "workload": "the headline of this line (one GPU's share of BASELINE configs[4])"}
        for nb_ in (16, 32, 64):
            if nb_ == B:
                continue
            try:
                reps = -(-nb_ // B)
                Ib0 = torch.cat([torch.roll(I0, 8 * k, 1) for k in range(reps)], 0)[:nb_].contiguous()
                Ib1 = torch.cat([torch.roll(I1, 8 * k, 1) for k in range(reps)], 0)[:nb_].contiguous()
                Fb = torch.empty((nb_, H, W, 2), dtype=torch.float32, device=dev)
                hb = max(2, hs * B // nb_)
                eb, _, _, _ = run(args.iterations, args.epsilon, hb, 2, inputs=(Ib0, Ib1), out=Fb)   # two warm-up steps: the first one of a new batch size allocates
                var[f"batch{nb_}_pairs_per_step"] = {"pairs_per_s": nb_ * hb / eb}
                if nb_ == 64:
                    # BASELINE configs[4]: 512 pairs over 8 GPUs = 64 pairs per GPU and step -- a first-class figure of the line
                    out["configs4_per_gpu_share_64_pairs"] = {"pairs_per_s": nb_ * hb / eb, "pairs_per_step": nb_, "steps": hb,
                                                              "ms_per_step": 1e3 * eb / hb,
                                                              "workload": "one GPU's share of BASELINE configs[4] (512 pairs / 8 GPUs), "
                                                                          f"{W}x{H} CV_32FC1, iterations={args.iterations}, epsilon={args.epsilon}"}
                del Ib0, Ib1, Fb
            except Exception as e:
                var[f"batch{nb_}_pairs_per_step"] = {"error": repr(e)[:200]}
        # SURVEY 8d config 2 "also run CV_8UC1": the same batch as 8-bit frames (the class converts them to f32 once per calc)
        try:
            J0, J1 = (I0 * 255).round().clamp(0, 255).to(torch.uint8), (I1 * 255).round().clamp(0, 255).to(torch.uint8)
            e8, _, _, _ = run(args.iterations, args.epsilon, hs, 1, inputs=(J0, J1))
            var["u8_input"] = {"pairs_per_s": B * hs / e8}
            del J0, J1
        except Exception as e:
            var["u8_input"] = {"error": repr(e)[:200]}
        # the reference's own calling pattern: one pair per calc() (no batching), back to back on one stream
        try:
            a1 = create(args.iterations, args.epsilon)
            one = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
            a1.calc(I0[0], I1[0], one)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(8):
                a1.calc(I0[i % B], I1[i % B], one)
            torch.cuda.synchronize()
            var["single_pair_calc_sequential"] = {"pairs_per_s": 8 / (time.perf_counter() - t1)}
            del a1, one
        except Exception as e:
            var["single_pair_calc_sequential"] = {"error": repr(e)[:200]}
        # the reference's own concurrency pattern (cudaoptflow/test/test_optflow.cpp:468-527): 16 objects, 16 streams, one calc() per
        # object and round, all in flight together -- what an UNCHANGED caller uses to get throughput
        for (it_, eps_, tag, fbk) in ((args.iterations, args.epsilon, "16_handles_16_streams_calc", {}),
                                      (10, 0.01, "16_handles_16_streams_calc_iterations10_eps0.01", {}),
                                      (300, 0.01, "16_handles_16_streams_calc_class_defaults", {}),
                                      # host_feedback = -1: nothing waits inside calc(), the 16 calcs of a round are all in flight (r02 behaviour)
                                      (300, 0.01, "16_handles_16_streams_calc_class_defaults_no_host_feedback", {"hostFeedback": -1})):
            try:
                nh = 16
                hs_ = [create(it_, eps_, **fbk) for _ in range(nh)]
                sts = [torch.cuda.Stream(device=dev) for _ in range(nh)]
                outs = [torch.empty((H, W, 2), dtype=torch.float32, device=dev) for _ in range(nh)]
                torch.cuda.synchronize()
                for k in range(nh):
                    hs_[k].calc(I0[k % B], I1[k % B], outs[k], stream=sts[k].cuda_stream)
                torch.cuda.synchronize()
                rounds = 2
                t1 = time.perf_counter()
                for _ in range(rounds):
                    for k in range(nh):
                        hs_[k].calc(I0[k % B], I1[k % B], outs[k], stream=sts[k].cuda_stream)
                torch.cuda.synchronize()
                e16 = time.perf_counter() - t1
                lone = create(it_, eps_, **fbk)
                chk = lone.calc(I0[3 % B], I1[3 % B])
                torch.cuda.synchronize()
                var[tag] = {"pairs_per_s": nh * rounds / e16, "handles": nh, "streams": nh, "iterations": it_, "epsilon": eps_,
                            "equals_lone_calc": bool(torch.equal(outs[3], chk))}
                del hs_, sts, outs, lone, chk
            except Exception as e:
                var[tag] = {"error": repr(e)[:200]}
        # ... and the pattern as the reference's test actually runs it: cv::parallel_for_ over the 16 (object, stream) pairs, i.e. 16 host
        # THREADS, each calc() followed by stream.waitForCompletion() in its own thread (test_optflow.cpp:474-484).  A wait inside one
        # thread's calc() (host feedback) holds up nobody else.  ctypes releases the GIL for the duration of the C call.
        for (it_, eps_, tag) in ((10, 0.01, "16_threads_16_handles_16_streams_calc_iterations10_eps0.01"),
                                 (300, 0.01, "16_threads_16_handles_16_streams_calc_class_defaults")):
            try:
                import threading
                nh, rounds = 16, 3
                hs_ = [create(it_, eps_) for _ in range(nh)]
                sts = [torch.cuda.Stream(device=dev) for _ in range(nh)]
                outs = [torch.empty((H, W, 2), dtype=torch.float32, device=dev) for _ in range(nh)]
                errs = []
                gate = threading.Barrier(nh + 1)

                def worker(k):
                    try:
                        torch.cuda.set_device(dev)
                        hs_[k].calc(I0[k % B], I1[k % B], outs[k], stream=sts[k].cuda_stream)   # warm-up
                        sts[k].synchronize()
                        gate.wait()
                        for _ in range(rounds):
                            hs_[k].calc(I0[k % B], I1[k % B], outs[k], stream=sts[k].cuda_stream)
                            sts[k].synchronize()
                        gate.wait()
                    except Exception as e:   # never leave the main thread at the barrier
                        errs.append(repr(e)[:200])
                        try:
                            gate.abort()
                        except Exception:
                            pass

                ths = [threading.Thread(target=worker, args=(k,)) for k in range(nh)]
                for t_ in ths:
                    t_.start()
                gate.wait(timeout=300)
                t1 = time.perf_counter()
                gate.wait(timeout=300)
                e16 = time.perf_counter() - t1
                for t_ in ths:
                    t_.join(timeout=60)
                torch.cuda.synchronize()
                lone = create(it_, eps_)
                chk = lone.calc(I0[3 % B], I1[3 % B])
                torch.cuda.synchronize()
                var[tag] = {"pairs_per_s": nh * rounds / e16, "handles": nh, "streams": nh, "host_threads": nh, "iterations": it_, "epsilon": eps_,
                            "equals_lone_calc": bool(torch.equal(outs[3], chk)), "errors": errs[:2]}
                del hs_, sts, outs, lone, chk
            except Exception as e:
                var[tag] = {"error": repr(e)[:200]}
        # the reference's own perf test of the class (cudaoptflow/perf/perf_optflow.cpp:283-311): ONE pair per calc(), class defaults
        # (300 iterations, epsilon 0.01), a 640 x 480 frame pair -- and the same at 1080p
        for (ww, hh, tag) in ((640, 480, "reference_perf_test_scenario_640x480_class_defaults_single_calc"), (W, H, "class_defaults_single_pair_calc_sequential")):
            try:
                # the same pair again and again, as the reference's TEST_CYCLE does -- since round 4 that is also the best case of the
                # handle's block-length history (exact); `..._three_scenes_in_turn` cycles three DIFFERENT synthetic scenes through the
                # handle, so every estimate comes from another scene (consecutive frames of a video lie between the two)
                q0, q1, _ = make_inputs(3, hh, ww, dev, distinct=3)
                a1 = cuda.OpticalFlowDual_TVL1.create()
                one = torch.empty((hh, ww, 2), dtype=torch.float32, device=dev)
                for _ in range(2):
                    a1.calc(q0[0], q1[0], one)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(8):
                    a1.calc(q0[0], q1[0], one)
                torch.cuda.synchronize()
                var[tag] = {"calcs_per_s": 8 / (time.perf_counter() - t1), "executed_iterations_per_warp_mean": float(np.mean(a1.lastIterations(0)))}
                # the same eight calcs timed ONE BY ONE (synchronised): a rare long calc -- two records of the pool's boxes show one of 20-70 ms
                # among eight of 1.6-2.3 ms (profiles/r18z, profiles/r19y), none in 4 200 calcs of tools/stall_probe.py -- then shows as
                # `per_calc_ms_max` instead of silently dividing the rate by five
                pc = []
                for i in range(8):
                    t2 = time.perf_counter()
                    a1.calc(q0[0], q1[0], one)
                    torch.cuda.synchronize()
                    pc.append(1e3 * (time.perf_counter() - t2))
                var[tag]["per_calc_ms_median"] = float(np.median(pc))
                var[tag]["per_calc_ms_max"] = float(np.max(pc))
                var[tag]["calcs_per_s_from_median"] = 1e3 / float(np.median(pc))
                for i in range(3):
                    a1.calc(q0[i], q1[i], one)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(9):
                    a1.calc(q0[i % 3], q1[i % 3], one)
                torch.cuda.synchronize()
                var[tag]["calcs_per_s_three_scenes_in_turn"] = 9 / (time.perf_counter() - t1)
                del a1, one, q0, q1
            except Exception as e:
                var[tag] = {"error": repr(e)[:200]}
        # The convergence-checked path as a VIDEO caller uses it: 17 consecutive frames of one synthetic sequence
        # (a texture drifting along a smooth flow field: frame k = the texture displaced by k x the field), ONE handle, class defaults,
        # one pair (frame k, frame k + 1) per calc().  The handle's block-length history then comes from the PREVIOUS pair of the same
        # scene -- between the repeat-same-pair figure (history exact) and three-scenes-in-turn (history from another scene) above.
        for (ww, hh, tag) in ((640, 480, "video_sequence_640x480_class_defaults"), (W, H, "video_sequence_class_defaults")):
            try:
                frames = synth_sequence(17, hh, ww, dev)
                a1 = cuda.OpticalFlowDual_TVL1.create()
                one = torch.empty((hh, ww, 2), dtype=torch.float32, device=dev)
                for _ in range(2):
                    a1.calc(frames[0], frames[1], one)
                torch.cuda.synchronize()
                its_ = []
                t1 = time.perf_counter()
                for k in range(1, 16):
                    a1.calc(frames[k], frames[k + 1], one)
                torch.cuda.synchronize()
                e_ = time.perf_counter() - t1
                its_ = float(np.mean(a1.lastIterations(0)))
                var[tag] = {"calcs_per_s": 15 / e_, "frames": 17, "executed_iterations_per_warp_mean_last_pair": its_,
                            "mean_flow_px_last_pair": float(one.abs().mean().item())}
                del a1, one, frames
            except Exception as e:
                var[tag] = {"error": repr(e)[:200]}
        # north_star "1080p/4K pairs": the same object on 3840x2160 pairs (B / 4 pairs per step = the pixels of the 1080p batch)
        try:
            # same motion in pixels as the 1080p pairs (flow_scale 3, texture sigma 6): five 0.8-scales cover it at either size
            n4 = max(4, B // 4)   # the pixels of B 1080p pairs
            K0, K1, base4k = make_inputs(n4, 2160, 3840, dev, distinct=1, flow_scale=3.0, sigma=6.0)
            F4 = torch.empty((n4, 2160, 3840, 2), dtype=torch.float32, device=dev)
            e4, _, _, _ = run(args.iterations, args.epsilon, hs, 1, inputs=(K0, K1), out=F4)
            ab4 = algo_bytes_per_pair(3840, 2160, warps, args.iterations)
            var["tvl1_4k_3840x2160"] = {"pairs_per_s": n4 * hs / e4, "batch": n4, "algorithmic_GB_per_pair": ab4 / 1e9,
                                        "megapixels_per_s": n4 * hs / e4 * 3840 * 2160 / 1e6,
                                        "epe_vs_analytic_flow_px": float(synth.epe(F4[0].cpu().numpy()[80:-80, 80:-80], base4k[0][2][80:-80, 80:-80]))}
            del K0, K1, F4
        except Exception as e:
            var["tvl1_4k_3840x2160"] = {"error": repr(e)[:200]}
        out["variants"] = var

    if rank == 0 and world == 1 and not args.no_cpu:
        # CPU baseline on this box's host cores.  kind "reference": the reference's OWN class cv::optflow::DualTVL1OpticalFlow --
        # modules/optflow/src/tvl1flow.cpp compiled verbatim against a stub core (oracle/_ref/libref_cpu.so, oracle/Makefile.ref;
        # parallel_for_ over OpenMP stripes like the library's, cv::remap / cv::resize from oracle/imgproc_ref.c) -- when that
        # prebuilt library travelled with the tree and the run uses the CPU class's arithmetic; otherwise kind "port": the oracle
        # (bit-identical to that class, tests/test_ref_pin.py).  Either way its per-pixel loops are the reference's, including the
        # SERIAL float error sum of estimateU: the GPU / CPU ratio says nothing about kernel quality (roofline.frac does).
        from oracle import oracle as O
        from oracle import refocl
        cit = args.cpu_iterations or (args.iterations if args.epsilon == 0 else 300)
        use_ref = int(P.semantics) == 0 and os.path.exists(refocl.cpu_lib_path())
        p = O.tvl1_params(iterations=cit, epsilon=args.epsilon, semantics=int(P.semantics))

        def cpu_calc(a, b):
            if use_ref:
                return refocl.cpu_tvl1_calc(a, b, inner_iterations=1, outer_iterations=cit, median_filtering=1, epsilon=args.epsilon)[0]
            return O.tvl1_calc(a, b, p)

        cb = timed_cpu_baseline(cpu_calc, base)
        ref0, times, med, best_nt, phys, sweep, one_core = (cb[k] for k in ("ref0", "times", "median_s", "threads", "physical_cores", "sweep", "one_core"))
        if ref0 is not None and cit == args.iterations:
            # BASELINE.json metric: "... EPE vs CPU ref" -- pair 0 of the timed batch against the CPU reference with the same
            # parameters; |1 - CCORR| is the reference's own comparator (cudaoptflow/test/test_optflow.cpp:465, 4e-3 there)
            out["epe_vs_cpu_ref_px"] = float(synth.epe(f0, ref0))
            out["ccorr_dissimilarity_vs_cpu_ref"] = float(max(synth.ccorr_dissimilarity(f0[..., 0], ref0[..., 0]),
                                                              synth.ccorr_dissimilarity(f0[..., 1], ref0[..., 1])))
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "pairs/s", "cores": best_nt, "kind": "reference" if use_ref else "port",
                               "sample": f"{len(times)} repetitions over {min(len(times), len(base))} distinct pair(s) {W}x{H} CV_32FC1 after one warm-up, "
                                         f"iterations={cit}, epsilon={args.epsilon}; median {med:.2f} s, min {min(times):.2f} s per pair; "
                                         + ("cv::optflow::DualTVL1OpticalFlow (tvl1flow.cpp verbatim, stub core, OpenMP stripes)" if use_ref
                                            else "oracle/tvl1_ref.c (OpenMP rows)"),
                               # the reference's class runs on OpenCV's parallel_for_; here its stripes run on the stub core's scheduler
                               # (oracle/refshim/cvstub: OpenMP static stripes) -- `cores` is the best thread count of the sweep below.
                               # `cpu_quota_cores` is what the container may use: the pool's GPU boxes give it 16 of the host's 128
                               # cores, so the sweep turns over at 16 threads (more are throttled) -- `cores` IS the available machine
                               "scheduler": "stub parallel_for_ (OpenMP static stripes), best of the thread sweep",
                               "cpu_quota_cores": cpu_quota_cores(),
                               "value_at_min": 1.0 / min(times), "physical_cores": phys, "logical_cpus": os.cpu_count(),
                               "thread_sweep_s_per_pair": {str(k): v for k, v in sweep.items()}, "one_core": one_core}
    if rank == 0 and world == 1 and not args.no_secondary:
        del I0, I1, flows
        torch.cuda.empty_cache()
        out["secondary"] = secondary(args)
    # The JSON line must be the LAST thing on stdout: RCCL writes a version banner to the C-level stdout, which stays in the C
    # buffer until the process exits when stdout is a pipe.  Tear the group down first, push the C buffer out, then print.
    if dist is not None and not exchange_hung:
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        emit(out)
    if exchange_hung:
        os._exit(0)


if __name__ == "__main__":
    main()
