"""GPU micro-sweep: cost of T temporally blocked TV-L1 iterations per HBM pass (tvl1_tb_kernels.hip).

For each time block T the full 1080p batched calc is timed at iterations = T and 2T (epsilon = 0);
the difference isolates 5 warps x T iterations over the 5-level pyramid, reported as G px-iterations/s
and as the per-pair pairs/s an N=10 run would reach at that slope.  Run on the GPU box:
    python tools/sweep_tb.py [--batch 16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--blocks", type=str, default="1,2,3,4,5,6,8,10")
    ap.add_argument("--tag", type=str, default="")
    ap.add_argument("--no-v1", action="store_true")
    args = ap.parse_args()
    os.environ["MIFLOW_TB_FORCE"] = "1"   # time blocks of exactly T (no cost-model decomposition)
    import torch
    from opencv_contrib_amd import cuda
    from bench import make_inputs, level_pixels
    dev = torch.device("cuda:0")
    W, H, B = 1920, 1080, args.batch
    I0, I1, _ = make_inputs(B, H, W, dev)
    flows = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
    px = sum(level_pixels(W, H))

    def t(iters, tb, exact=False):
        alg = cuda.OpticalFlowDual_TVL1.create(iterations=iters, epsilon=0.0, exactMath=exact, timeBlock=tb)
        alg.calc_batch(I0, I1, flows)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):   # median of individually timed calls (the slope of two such timings amplifies noise)
            t0 = time.perf_counter()
            alg.calc_batch(I0, I1, flows)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    res = {}
    t0 = t(0, 1)
    res["warp_only_ms_per_pair"] = 1e3 * t0 / B
    for T in [int(x) for x in args.blocks.split(",")]:
        a, b = t(T, T), t(2 * T, T)
        per_iter = (b - a) / T / B          # s per (pair, iteration) over all levels and 5 warps
        res[f"T{T}"] = {"ms_N=T": 1e3 * a / B, "ms_N=2T": 1e3 * b / B,
                        "Gpxiter_per_s": px * 5 / per_iter / 1e9,
                        "pairs_per_s_at_N10_extrapolated": 1.0 / (t0 / B + 10 * per_iter)}
    if not args.no_v1:
        a, b = t(2, 1, exact=True), t(4, 1, exact=True)
        res["v1_exact"] = {"Gpxiter_per_s": px * 5 / ((b - a) / 2 / B) / 1e9}
    res["tag"] = args.tag
    res["env"] = {k: v for k, v in os.environ.items() if k.startswith("MIFLOW_")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
