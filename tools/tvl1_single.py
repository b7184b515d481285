"""One pair per calc() with class defaults (cudaoptflow/perf/perf_optflow.cpp:283-311), n calcs on one handle: calcs per second.
usage: python tools/tvl1_single.py <w> <h> [n]      (trace it: rocprofv3 --kernel-trace ... then tools/trace_summary.py <dir> k_convert)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from opencv_contrib_amd import cuda, synth

w, h = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
I0, I1, _ = synth.flow_pair(h, w, seed=1234)
a, b = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
alg = cuda.OpticalFlowDual_TVL1.create()
out = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
for _ in range(3):
    alg.calc(a, b, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    alg.calc(a, b, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{w}x{h} class defaults, one pair per calc: {1.0 / dt:.1f} calcs/s ({dt * 1e3:.3f} ms); iterations {alg.lastIterations(0)}")
