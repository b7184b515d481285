"""The batched Farneback calc of bench.py alone (32 copies of one 640 x 480 pair, class defaults), for profiler passes.
usage: python tools/fb_batch.py [pairs] [reps] [width height]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (640, 480)
P = [synth.flow_pair(H, W, seed=1234 + k, dtype="u8") for k in range(min(nb, 4))]   # four distinct scenes in turn
b0 = [torch.from_numpy(P[i % len(P)][0]).to(dev) for i in range(nb)]
b1 = [torch.from_numpy(P[i % len(P)][1]).to(dev) for i in range(nb)]
flows = torch.empty((nb, H, W, 2), dtype=torch.float32, device=dev)
alg = cuda.FarnebackOpticalFlow.create()
alg.calc_batch(b0, b1, flows)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    alg.calc_batch(b0, b1, flows)
torch.cuda.synchronize()
rate = nb * reps / (time.perf_counter() - t)
import hashlib
print(f"batched {nb} x {W}x{H}: digest {hashlib.sha256(flows.cpu().numpy().tobytes()).hexdigest()[:12]} {rate:.1f} pairs/s", flush=True)
