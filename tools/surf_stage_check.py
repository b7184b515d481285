"""Digest of SURF descriptors (64 and 128 floats) of the 4K blob frame and of a frame with provided keypoints of many sizes: run once with
MIFLOW_SURF_STAGE_S=0 (every patch through the global-memory kernel) and once with the default (large features through the LDS-staged
kernel) -- the two must print the same lines (the staged kernel reads the same texels and adds them in the same order)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda", 0)
dg = lambda t: hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()[:16]
img = torch.from_numpy(synth.blob_image(2160, 3840, seed=7)).to(dev)
for ext in (False, True):
    kp, d = cuda.SURF_CUDA.create(400.0, 4, 2, ext).detectWithDescriptors(img)
    print("4K blob extended", ext, kp.shape[1], dg(kp), dg(d), flush=True)
img2 = torch.from_numpy(np.rint(synth.texture(600, 800, 11, 2.0)).astype(np.uint8)).to(dev)
rng = np.random.default_rng(5)
n = 300
kp = np.zeros((7, n), np.float32)
kp[0] = rng.uniform(1, 799, n); kp[1] = rng.uniform(1, 599, n)
kp.view(np.int32)[2] = 1
kp[4] = rng.choice([4, 9, 30, 44, 46, 60, 90, 133, 216, 300, 420, 700], n); kp[5] = rng.uniform(0, 360, n)
for ext in (False, True):
    up = cuda.SURF_CUDA.create(400.0, 4, 2, ext, 0.01, True)
    k2, d = up.detectWithDescriptors(img2, keypoints=torch.from_numpy(kp).to(dev), useProvidedKeypoints=True)
    print("provided sizes 4..700 extended", ext, dg(d), bool(torch.isfinite(d).all()), flush=True)
