"""gamma != 0 (the illumination channel): pairs per second of the blocked path (k_iterate_tbr GAM, round 6) against one launch per
iteration (timeBlock = 1, the only fast-math path before round 6) and against gamma = 0, at 1080p.
usage: python tools/gamma_bench.py [pairs] [steps] [only: substring of a case name, e.g. "blocked" for profiler runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from opencv_contrib_amd import cuda, synth

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
only = sys.argv[3] if len(sys.argv) > 3 else ""
h, w = 1080, 1920
pairs = [synth.flow_pair(h, w, seed=1234 + k)[:2] for k in range(4)]
I0 = torch.stack([torch.from_numpy(pairs[k % 4][0]) for k in range(n)]).to(dev)
I1 = torch.stack([torch.from_numpy(np.clip(pairs[k % 4][1] * 1.05 + 0.02, 0, 1).astype(np.float32)) for k in range(n)]).to(dev)
out = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
for name, kw in (("gamma0 N=10", dict(iterations=10, epsilon=0.0)),
                 ("gamma1 N=10 blocked", dict(iterations=10, epsilon=0.0, gamma=1.0)),
                 ("gamma1 N=10 timeBlock=5", dict(iterations=10, epsilon=0.0, gamma=1.0, timeBlock=5)),
                 ("gamma1 N=10 one launch per iteration", dict(iterations=10, epsilon=0.0, gamma=1.0, timeBlock=1)),
                 ("gamma1 N=10 cuda semantics", dict(iterations=10, epsilon=0.0, gamma=1.0, semantics=1)),
                 ("gamma0 class defaults", dict()),
                 ("gamma1 class defaults", dict(gamma=1.0))):
    if only and only not in name:
        continue
    alg = cuda.OpticalFlowDual_TVL1.create(**kw)
    for _ in range(2):
        alg.calc_batch(I0, I1, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        alg.calc_batch(I0, I1, out)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name}: {n * steps / el:.0f} pairs/s  (iterations per warp, pair 0: {alg.lastIterations(0)})", flush=True)
    del alg
