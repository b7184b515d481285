"""A few class-default TV-L1 calcs of ONE 640x480 pair on one handle (the reference's perf scenario, perf_optflow.cpp:283-311), for
`rocprofv3 --kernel-trace`: tools/timeline.py prints the last calc's launches.  usage: python tools/single_calc_trace.py [W H [n]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
frames = [torch.from_numpy(synth.flow_pair(H, W, seed=1234 + k)[0]).to(dev) for k in range(3)]
I0, I1, _ = synth.flow_pair(H, W, seed=1234)
a, b = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
alg = cuda.OpticalFlowDual_TVL1.create()
out = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
for _ in range(3):
    alg.calc(a, b, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    alg.calc(a, b, out)
torch.cuda.synchronize()
print(f"{W}x{H} class defaults, one pair per calc: {n / (time.perf_counter() - t0):.1f} calcs/s; iterations {alg.lastIterations(0)}", flush=True)
