"""EPE against the CPU-class oracle of a 1080p batch through the blocked fixed-work path, for the p-storage experiment
(MIFLOW_TB_P16=0/1 in the environment; MIFLOW_TILE_MAXPX=0 keeps every level on the streaming kernel).  usage: python tools/p16_probe.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opencv_contrib_amd import cuda, synth
from oracle import oracle as O
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pairs = [synth.flow_pair(1080, 1920, seed=1234 + i) for i in range(min(B, 4))]
I0 = torch.stack([torch.from_numpy(pairs[i % len(pairs)][0]) for i in range(B)]).to(dev)
I1 = torch.stack([torch.from_numpy(pairs[i % len(pairs)][1]) for i in range(B)]).to(dev)
alg = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0)
F = alg.calc_batch(I0, I1)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    alg.calc_batch(I0, I1, F)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
f = F.cpu().numpy()
ep = []
for i in range(min(2, len(pairs))):
    ref = O.tvl1_calc(pairs[i][0], pairs[i][1], O.tvl1_params(iterations=10, epsilon=0.0))
    d = np.sqrt(((f[i] - ref) ** 2).sum(-1))
    ep.append((float(d.mean()), float(np.percentile(d, 99)), float(d.max()), float(np.sqrt(((f[i] - pairs[i][2]) ** 2).sum(-1))[40:-40, 40:-40].mean())))
print(f"P16={os.environ.get('MIFLOW_TB_P16','0')} TILE_MAXPX={os.environ.get('MIFLOW_TILE_MAXPX','default')} B={B}: {B / dt:.1f} pairs/s; (mean EPE, p99, max, EPE vs analytic) per pair:", ep, flush=True)
