"""Descriptor time of SURF_CUDA on the 4K blob frame by octave: the keypoints of a 4-octave detect are split by their octave row and each
subset is described alone (useProvidedKeypoints with the detected orientation kept: upright handle, angles already in the matrix)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda", 0)
img = torch.from_numpy(synth.blob_image(2160, 3840, seed=7)).to(dev)
surf = cuda.SURF_CUDA.create(400.0)
kp = surf.detect(img).clone()
torch.cuda.synchronize()
up = cuda.SURF_CUDA.create(400.0, 4, 2, False, 0.01, True)   # upright: detectWithDescriptors(useProvidedKeypoints) leaves the angle row alone
octs = kp[3].view(torch.int32)
for sel, name in [(octs >= 0, "all")] + [(octs == o, f"octave {o}") for o in range(4)]:
    sub = kp[:, sel].contiguous()
    if sub.shape[1] == 0:
        continue
    for _ in range(2):
        up.detectWithDescriptors(img, keypoints=sub, useProvidedKeypoints=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        up.detectWithDescriptors(img, keypoints=sub, useProvidedKeypoints=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {sub.shape[1]} features, mean size {float(sub[4].mean()):.1f} px, {1e6 * dt:.0f} us per describe, {1e9 * dt / sub.shape[1]:.0f} ns per feature", flush=True)
