"""Fused warp (MIFLOW_TB_FW) against the separate warp launch: the flows of fixed-work calcs must be bit-identical.
usage: python tools/fw_check.py            (parent: runs itself with MIFLOW_TB_FW=0 and =1 and compares)
       python tools/fw_check.py child OUT  (one process: writes the flows of every case to OUT.npz)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [  # (h, w, pairs, iterations, semantics, dtype)
    (1080, 1920, 8, 10, 0, "f32"), (1080, 1920, 8, 10, 1, "f32"), (480, 640, 16, 10, 0, "u8"), (388, 584, 12, 10, 0, "f32"),
    (217, 333, 40, 10, 0, "f32"), (1080, 1920, 8, 20, 0, "f32"), (720, 1280, 8, 10, 0, "f32"), (1080, 1920, 64, 10, 0, "f32"),
]


def child(out):
    import numpy as np
    import torch
    from opencv_contrib_amd import cuda, synth
    dev = torch.device("cuda:0")
    res = {}
    for i, (h, w, n, it, sem, dt) in enumerate(CASES):
        pairs = [synth.flow_pair(h, w, seed=900 + 7 * i + k, dtype=dt) for k in range(min(n, 4))]
        I0 = torch.stack([torch.from_numpy(pairs[k % len(pairs)][0]) for k in range(n)]).to(dev)
        I1 = torch.stack([torch.from_numpy(pairs[k % len(pairs)][1]) for k in range(n)]).to(dev)
        alg = cuda.OpticalFlowDual_TVL1.create(iterations=it, epsilon=0.0, semantics=sem)
        F = alg.calc_batch(I0, I1)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 3
        for _ in range(reps):
            alg.calc_batch(I0, I1, F)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t) / reps
        f = F.cpu().numpy()
        res[f"c{i}"] = f[: min(n, 4)]
        print(f"case {i} {w}x{h} x{n} it={it} sem={sem} {dt}: {n / dt_:.1f} pairs/s  finite={np.isfinite(f).all()} "
              f"epe_gt={synth.epe(f[0], pairs[0][2]):.4f}", flush=True)
    np.savez(out, **res)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        return child(sys.argv[2])
    import numpy as np
    outs = {}
    for fw in ("0", "1"):
        out = f"/tmp/fw_check_{fw}.npz"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", out], env=dict(os.environ, MIFLOW_TB_FW=fw, MIFLOW_TB_VERBOSE="1" if fw == "1" else ""),
                           capture_output=True, text=True, timeout=900)
        print(f"--- MIFLOW_TB_FW={fw} rc={r.returncode}\n{r.stdout}")
        if fw == "1":
            print("\n".join([l for l in r.stderr.splitlines() if "fused" in l][:6]))
        if r.returncode != 0:
            print(r.stderr[-3000:])
            return 1
        outs[fw] = np.load(out)
    bad = 0
    for k in outs["0"].files:
        a, b = outs["0"][k], outs["1"][k]
        same = np.array_equal(a, b)
        d = np.abs(a - b)
        print(k, "IDENTICAL" if same else f"DIFFERENT: max {d.max():.3e}, mean {d.mean():.3e}, differing px {(d.max(-1) > 0).mean():.4f}")
        if not same:
            bad += 1
            idx = np.argwhere(d.max(-1) > 0)
            print("   first differing (pair, y, x):", idx[:5].tolist(), " y range", idx[:, 1].min(), idx[:, 1].max(), " x range", idx[:, 2].min(), idx[:, 2].max())
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
