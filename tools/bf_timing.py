"""GPU session helper: __graft_entry__.smoke() and a timing of the brute-force matcher shapes (one process = one torch import)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
from opencv_contrib_amd import cuda  # noqa: E402


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    g.smoke()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    out = {}
    for d in (64, 128):
        x = torch.from_numpy(rng.standard_normal((12564, d)).astype(np.float32)).to(dev)
        for w in ("1", "4"):
            os.environ["MIFLOW_BF_W"] = w
            for norm, nn in ((4, "L2"), (2, "L1")):
                m = cuda.createBFMatcher(norm)
                out[f"knn2_{nn}_d{d}_w{w}_ms"] = timed(lambda: m.knnMatchDevice(x, x, k=2))
        os.environ.pop("MIFLOW_BF_W", None)
        m = cuda.createBFMatcher(4)
        out[f"knn8_L2_d{d}_ms"] = timed(lambda: m.knnMatchDevice(x, x, k=8))
        out[f"radius_L2_d{d}_ms"] = timed(lambda: m.radiusMatchDevice(x, x, 0.5), reps=2)
    x = torch.from_numpy(rng.standard_normal((4096, 304)).astype(np.float32)).to(dev)
    out["knn2_L2_d304_4096_ms"] = timed(lambda: cuda.createBFMatcher(4).knnMatchDevice(x, x, k=2), reps=2)
    print(json.dumps({k: round(v, 3) for k, v in out.items()}))


if __name__ == "__main__":
    main()
