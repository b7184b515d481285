"""Latency spikes of one-pair calc()s: times N calcs one by one (synchronised) and reports the outliers.  usage: python tools/stall_probe.py [W H N]
Each configuration: class defaults with the polled host feedback (default), without it (hostFeedback=-1), and fixed work (N = 10, eps 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from opencv_contrib_amd import cuda, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
dev = torch.device("cuda:0")
I0, I1, _ = synth.flow_pair(H, W, seed=1234)
a, b = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
out = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
for name, kw in (("class defaults, polled feedback", dict()), ("class defaults, no host feedback", dict(hostFeedback=-1)), ("N=10 eps=0", dict(iterations=10, epsilon=0.0))):
    alg = cuda.OpticalFlowDual_TVL1.create(**kw)
    for _ in range(5): alg.calc(a, b, out)
    torch.cuda.synchronize()
    ts = np.empty(N)
    for i in range(N):
        t = time.perf_counter(); alg.calc(a, b, out); torch.cuda.synchronize(); ts[i] = 1e3 * (time.perf_counter() - t)
    med = np.median(ts)
    big = ts[ts > 3 * med]
    print(f"{W}x{H} {name}: median {med:.3f} ms, p99 {np.percentile(ts, 99):.3f}, max {ts.max():.2f}; {len(big)} of {N} calcs above 3x the median: {[round(x, 1) for x in sorted(big)[-8:]]}", flush=True)
