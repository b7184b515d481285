"""The batched Farneback calc of bench.py alone (32 copies of one 640 x 480 pair, class defaults), for profiler passes.
usage: python tools/fb_batch.py [pairs] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
I0, I1, _ = synth.flow_pair(480, 640, seed=1234, dtype="u8")
b0 = [torch.from_numpy(I0).to(dev) for _ in range(nb)]
b1 = [torch.from_numpy(I1).to(dev) for _ in range(nb)]
flows = torch.empty((nb, 480, 640, 2), dtype=torch.float32, device=dev)
alg = cuda.FarnebackOpticalFlow.create()
alg.calc_batch(b0, b1, flows)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    alg.calc_batch(b0, b1, flows)
torch.cuda.synchronize()
print(f"batched {nb}: {nb * reps / (time.perf_counter() - t):.1f} pairs/s", flush=True)
