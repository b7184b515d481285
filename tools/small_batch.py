"""Batches of small frames and the coarse levels of larger ones (the register-tile kernel's territory): pairs per second, fixed work
(N = 10, epsilon 0) and class defaults.  usage: python tools/small_batch.py [cd | fixed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from opencv_contrib_amd import cuda, synth

dev = torch.device("cuda", 0)
for (w, h, n) in ((320, 240, 16), (320, 240, 64), (640, 480, 16), (1280, 720, 8), (1920, 1080, 4)):
    pairs = [synth.flow_pair(h, w, seed=1234 + k)[:2] for k in range(min(n, 4))]
    I0 = torch.stack([torch.from_numpy(pairs[k % len(pairs)][0]) for k in range(n)]).to(dev)
    I1 = torch.stack([torch.from_numpy(pairs[k % len(pairs)][1]) for k in range(n)]).to(dev)
    out = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
    res = []
    modes = ((10, 0.0), (300, 0.01)) if len(sys.argv) < 2 else (((300, 0.01),) if sys.argv[1] == "cd" else ((10, 0.0),))
    for (it, eps) in modes:
        alg = cuda.OpticalFlowDual_TVL1.create(iterations=it, epsilon=eps)
        for _ in range(3):
            alg.calc_batch(I0, I1, out)
        torch.cuda.synchronize()
        steps = 10
        t0 = time.perf_counter()
        for _ in range(steps):
            alg.calc_batch(I0, I1, out)
        torch.cuda.synchronize()
        res.append(n * steps / (time.perf_counter() - t0))
    if len(res) == 2:
        print(f"{w}x{h} x {n} pairs: N=10 eps=0 {res[0]:.0f} pairs/s, class defaults {res[1]:.0f} pairs/s", flush=True)
    else:
        print(f"{w}x{h} x {n} pairs ({sys.argv[1]}): {res[0]:.0f} pairs/s", flush=True)
