"""Mean EPE of the fast-math TV-L1 path against the CPU-class oracle on the parity tests' inputs (prints; no assertions).
MIFLOW_WARP_FAST=0/1 etc. in the environment select the variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opencv_contrib_amd import cuda, synth
from oracle import oracle as O

dev = torch.device("cuda:0")
def run(I0, I1, **kw):
    alg = cuda.OpticalFlowDual_TVL1.create(**kw)
    f = alg.calc(torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev))
    torch.cuda.synchronize()
    return f.cpu().numpy()
def epe(a, b): return float(np.sqrt(((a - b) ** 2).sum(-1)).mean())

I0, I1, _ = synth.flow_pair(240, 320, seed=17)
I1g = np.clip(I1 * 1.08 + 0.02, 0, 1).astype(np.float32)
for sem in (0, 1):
    ref = O.tvl1_calc(I0, I1g, O.tvl1_params(iterations=10, epsilon=0.0, gamma=1.0, semantics=sem))
    print("gamma=1 sem", sem, "fast EPE", epe(run(I0, I1g, iterations=10, epsilon=0.0, gamma=1.0, semantics=sem, exactMath=False), ref),
          "exact EPE", epe(run(I0, I1g, iterations=10, epsilon=0.0, gamma=1.0, semantics=sem, exactMath=True), ref))
for (h, w, seed) in ((388, 584, 78), (240, 320, 17), (480, 640, 5)):
    I0, I1, _ = synth.flow_pair(h, w, seed=seed)
    ref = O.tvl1_calc(I0, I1, O.tvl1_params(iterations=10, epsilon=0.0))
    print(h, w, "N=10 fast EPE", epe(run(I0, I1, iterations=10, epsilon=0.0, exactMath=False), ref),
          "exact EPE", epe(run(I0, I1, iterations=10, epsilon=0.0, exactMath=True), ref))
