"""Digests of SURF keypoints + descriptors for a few frames / parameter sets (one process: the switches are read once).
usage: python tools/surf_digest.py"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda:0")
for (shape, thr, octaves, layers, ext, ratio, masked) in (((2160, 3840), 400.0, 4, 2, False, 0.01, False), ((300, 400), 100.0, 4, 2, False, 0.05, False),
                                                          ((720, 1283), 50.0, 3, 1, True, 0.05, True), ((480, 640), 200.0, 4, 2, False, 0.0005, False),
                                                          ((1080, 1920), 300.0, 2, 2, False, 0.02, True)):
    img = synth.blob_image(*shape, seed=17)
    m = None
    if masked:
        mm = np.zeros_like(img); mm[shape[0] // 7: shape[0] - 31, 45: shape[1] // 2 + 100] = 3
        m = torch.from_numpy(mm).to(dev)
    alg = cuda.SURF_CUDA.create(thr, octaves, layers, ext, ratio, False)
    kp, d = alg.detectWithDescriptors(torch.from_numpy(img).to(dev), m)
    h = hashlib.sha256(kp.cpu().numpy().tobytes() + d.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"surf {shape[1]}x{shape[0]} thr={thr} o={octaves} l={layers} ext={int(ext)} ratio={ratio} mask={int(masked)}: n={kp.shape[1]} digest {h}", flush=True)
