"""Detect-only time of SURF_CUDA on the 4K blob frame for 1..4 octaves (MIFLOW_SURF_FUSED=0: octave-by-octave launches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda", 0)
img = torch.from_numpy(synth.blob_image(2160, 3840, seed=7)).to(dev)
for no in (1, 2, 3, 4):
    surf = cuda.SURF_CUDA.create(400.0, no)
    for _ in range(3):
        kp = surf.detect(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        kp = surf.detect(img)
    torch.cuda.synchronize()
    print(f"fused={os.environ.get('MIFLOW_SURF_FUSED', '1')} octaves={no}: {1e6 * (time.perf_counter() - t0) / 20:.0f} us per detect, {kp.shape[1]} features", flush=True)
