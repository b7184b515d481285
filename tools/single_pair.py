"""Sequential single-pair calc() rate (the reference's own calling pattern: no batching) at a given size; tuning switches come
from the environment (read once by libmiflow.so).  usage: python tools/single_pair.py [W H [n]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
dev = torch.device("cuda:0")
I0, I1, _ = synth.flow_pair(H, W, seed=1234)
a, b = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
out = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
for defaults in (False, True):
    alg = cuda.OpticalFlowDual_TVL1.create(**({} if defaults else dict(iterations=10, epsilon=0.0)))
    for _ in range(3):
        alg.calc(a, b, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        alg.calc(a, b, out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{W}x{H} {'class defaults' if defaults else 'N=10 eps=0'}: {n / dt:.1f} calc/s ({1e3 * dt / n:.3f} ms)  "
          f"TILE_MAXPX={os.environ.get('MIFLOW_TILE_MAXPX', 'default')} VARIANT={os.environ.get('MIFLOW_TILE_VARIANT', 'default')}", flush=True)
    del alg
