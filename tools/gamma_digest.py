"""Digests of gamma != 0 calcs whose levels are small enough for the register-tile kernel: run once as is and once with MIFLOW_TILE_MAXPX=0
(every level on the streaming kernel) -- the two kernels carry the illumination channel with the same operations in the same order, so the
digests (and the iteration counts of the convergence-checked calcs) must be equal.  usage: python tools/gamma_digest.py"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opencv_contrib_amd import cuda, synth

dev = torch.device("cuda:0")


def dig(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]


for (h, w, seed) in ((300, 420, 5), (135, 240, 6), (480, 640, 7)):
    I0, I1, _ = synth.flow_pair(h, w, seed=seed)
    I1 = np.clip(I1 * 1.05 + 0.015, 0, 1).astype(np.float32)
    a, b = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
    for name, kw in (("N10", dict(iterations=10, epsilon=0.0, gamma=1.0)), ("N7", dict(iterations=7, epsilon=0.0, gamma=0.6)),
                     ("N23_cuda", dict(iterations=23, epsilon=0.0, gamma=1.0, semantics=1)), ("defaults", dict(gamma=0.8)),
                     ("defaults_cuda", dict(gamma=0.8, semantics=1))):
        alg = cuda.OpticalFlowDual_TVL1.create(**kw)
        f = alg.calc(a, b)
        torch.cuda.synchronize()
        print(f"gamma {w}x{h} {name}: digest {dig(f)} iterations {alg.lastIterations(0)}", flush=True)
        if name in ("N10", "defaults"):
            fb = alg.calc_batch([a, a.flip(0).contiguous(), a], [b, b.flip(0).contiguous(), b])
            torch.cuda.synchronize()
            print(f"gamma {w}x{h} {name} batch of 3: digest {dig(fb)} pair 0 equals the single calc: {bool(torch.equal(fb[0], f))}", flush=True)
