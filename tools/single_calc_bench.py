"""One pair per calc(), class defaults (the reference's own perf test, cudaoptflow/perf/perf_optflow.cpp:283-311) and the reference test's
literal setting, with mi_tvl1_params.host_feedback automatic (0) and off (-1): calcs per second, and that the flows are identical."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from opencv_contrib_amd import cuda, synth

dev = torch.device("cuda", 0)
for (h, w) in ((480, 640), (1080, 1920)):
    I0, I1, _ = synth.flow_pair(h, w, seed=1234)
    a, b = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
    for (it, eps, tag) in ((300, 0.01, "class defaults"), (10, 0.01, "N=10 eps=0.01")):
        res = {}
        for fb in (-1, 0):
            alg = cuda.OpticalFlowDual_TVL1.create(iterations=it, epsilon=eps, hostFeedback=fb)
            out = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
            for _ in range(3):
                alg.calc(a, b, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                alg.calc(a, b, out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            res[fb] = (out.clone(), 1.0 / dt, alg.lastIterations(0))
        same = bool(torch.equal(res[-1][0], res[0][0])) and res[-1][2] == res[0][2]
        print(f"{w}x{h} {tag}: host_feedback off {res[-1][1]:.1f} calcs/s, automatic {res[0][1]:.1f} calcs/s, identical flows and counts: {same}; "
              f"mean iterations {np.mean(res[0][2]):.1f}", flush=True)
