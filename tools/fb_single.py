"""Sequential single-pair Farneback calc() rate at 640x480 (BASELINE configs[0]'s shape: one pair per call).  usage: python tools/fb_single.py [W H [n]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
A0, A1, _ = synth.flow_pair(H, W, seed=5, dtype="u8")
a, b = torch.from_numpy(A0).to(dev), torch.from_numpy(A1).to(dev)
alg = cuda.FarnebackOpticalFlow.create()
out = alg.calc(a, b)
for _ in range(5):
    alg.calc(a, b, out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    alg.calc(a, b, out)
torch.cuda.synchronize()
dt = time.perf_counter() - t
import hashlib
print(f"farneback {W}x{H}: {n / dt:.1f} calc/s ({1e3 * dt / n:.3f} ms)  digest {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
