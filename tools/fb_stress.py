"""Random batched Farneback calls against single calc()s of the same pairs, bit for bit: odd sizes, 8-bit and float frames, pitched (ROI)
inputs, window / pyramid parameters, with whatever MIFLOW_FB_GROUP_MB the environment sets (a small budget cuts every level into many
pair groups on two streams).  usage: python tools/fb_stress.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n):
    h, w = int(rng.integers(70, 420)), int(rng.integers(70, 560))
    B = int(rng.integers(2, 41))
    f32 = bool(rng.integers(2))
    kw = dict(winSize=int((9, 13, 15, 21, 11)[int(rng.integers(5))]), numLevels=int(rng.integers(1, 6)), numIters=int(rng.integers(1, 5)),
              pyrScale=float((0.5, 0.5, 0.7)[int(rng.integers(3))]), flags=int((0, 256)[int(rng.integers(2))]), fastPyramids=False)
    if kw["pyrScale"] == 0.5 and rng.integers(4) == 0:
        kw["fastPyramids"] = True
    pitched = bool(rng.integers(2))
    scenes = [synth.flow_pair(h, w, seed=int(rng.integers(1, 10 ** 6)), dtype="f32" if f32 else "u8")[:2] for _ in range(min(B, 3))]
    def dev_mat(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        if not pitched:
            return t
        big = torch.zeros((h + 3, w + 37), dtype=t.dtype, device=dev)
        big[2:2 + h, 5:5 + w] = t
        return big[2:2 + h, 5:5 + w]          # a view with a row pitch: what a GpuMat ROI is
    I0 = [dev_mat(scenes[i % len(scenes)][0]) for i in range(B)]
    I1 = [dev_mat(scenes[i % len(scenes)][1]) for i in range(B)]
    alg, one = cuda.FarnebackOpticalFlow.create(**kw), cuda.FarnebackOpticalFlow.create(**kw)
    out = alg.calc_batch(I0, I1)
    singles = [one.calc(I0[i], I1[i]).clone() for i in range(min(B, 3))]
    torch.cuda.synchronize()
    ok = all(torch.equal(out[i], singles[i % len(singles)]) for i in range(B))
    bad += not ok
    print(f"case {case}: {w}x{h} x {B} {'f32' if f32 else 'u8'} pitched={pitched} {kw} -> {'ok' if ok else 'DIFFERENT'}", flush=True)
print("fb_stress:", "ok" if not bad else f"{bad} case(s) differ")
sys.exit(1 if bad else 0)
