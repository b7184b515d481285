"""20 x SURF_CUDA detect (no descriptors) and 10 x detect + describe on the 4K blob frame of BASELINE configs[3], one handle, one stream: the
input of a rocprofv3 --kernel-trace --stats run in which every kernel's average is a per-frame figure."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from opencv_contrib_amd import cuda, synth

dev = torch.device("cuda", 0)
img = torch.from_numpy(synth.blob_image(2160, 3840, seed=7)).to(dev)
surf = cuda.SURF_CUDA.create(400.0)
for _ in range(3):
    kp = surf.detect(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    kp = surf.detect(img)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(10):
    kp2, desc = surf.detectWithDescriptors(img)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"detect only {20 / (t1 - t0):.1f} frames/s, detect + describe {10 / (t2 - t1):.1f} frames/s, {kp.shape[1]} features")
