"""Digest + rate of a class-default (300 iterations, eps 0.01) 1080p batch: the speculative steps on the streaming kernel.  usage: python tools/defaults_digest.py [pairs] [both]   (both: also the fixed-work setting iterations=10, epsilon=0)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pairs = [synth.flow_pair(1080, 1920, seed=1234 + i) for i in range(min(B, 4))]
I0 = torch.stack([torch.from_numpy(pairs[i % len(pairs)][0]) for i in range(B)]).to(dev)
I1 = torch.stack([torch.from_numpy(pairs[i % len(pairs)][1]) for i in range(B)]).to(dev)
alg = cuda.OpticalFlowDual_TVL1.create()
F = alg.calc_batch(I0, I1)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    alg.calc_batch(I0, I1, F)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
print(f"class defaults 1080p x {B}: {B / dt:.1f} pairs/s digest {hashlib.sha256(F.cpu().numpy().tobytes()).hexdigest()[:16]} iterations {alg.lastIterations(0)[0][:5]}", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "both":
    a10 = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0)
    F10 = a10.calc_batch(I0, I1)
    torch.cuda.synchronize()
    print(f"iterations=10 epsilon=0 1080p x {B}: digest {hashlib.sha256(F10.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
