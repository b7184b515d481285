"""Python twin of samples/optical_flow.cpp: the four optical-flow classes of the cv2.cuda-style mirror on a synthetic pair.
Needs an MI355X (there is no CPU fallback):  python samples/optical_flow.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencv_contrib_amd import cuda, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    I0, I1, gt = synth.flow_pair(240, 320, seed=1234, dtype="u8")
    t0, t1 = torch.from_numpy(I0).to(dev), torch.from_numpy(I1).to(dev)
    for name, alg in (("TV-L1", cuda.OpticalFlowDual_TVL1.create()), ("Farneback", cuda.FarnebackOpticalFlow.create()),
                      ("DensePyrLK", cuda.DensePyrLKOpticalFlow.create())):
        flow = alg.calc(t0, t1).cpu().numpy()
        print(f"{name:12s} EPE vs the analytic flow {synth.epe(flow[20:-20, 20:-20], gt[20:-20, 20:-20]):.3f} px")
    ys, xs = np.mgrid[40:200:20, 40:280:20]
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)
    nxt, status, err = cuda.SparsePyrLKOpticalFlow.create().calc(t0, t1, torch.from_numpy(pts).to(dev))
    d = nxt.cpu().numpy()[0] - pts
    g = gt[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    ok = status.cpu().numpy()[0] > 0
    print(f"SparsePyrLK  {int(ok.sum())} / {len(pts)} points tracked, median error {np.median(np.hypot(*(d - g)[ok].T)):.3f} px")


if __name__ == "__main__":
    main()
