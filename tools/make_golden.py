"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference's own goldens (opencv_extra: RubberWhale, tvl1_flow.flo, aloe-disp*.png) are not
available in this environment and the reference cannot be built (no OpenCV core), so these
fixtures pin the ORACLE (regression) and give the GPU tests a file-based target; they are not
outputs of the reference binary.  Re-run: python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from opencv_contrib_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def tvl1():
    cases = {
        # name: (h, w, dtype, params)
        "tvl1_f32_96x128_it10": (96, 128, "f32", dict(iterations=10, epsilon=0.0)),
        "tvl1_u8_96x128_it10": (96, 128, "u8", dict(iterations=10, epsilon=0.0)),
        "tvl1_f32_77x101_eps": (77, 101, "f32", dict(iterations=300, epsilon=0.01)),
        "tvl1_f32_96x128_it10_cuda": (96, 128, "f32", dict(iterations=10, epsilon=0.0, semantics=1)),
    }
    for name, (h, w, dt, kw) in cases.items():
        I0, I1, gt = synth.flow_pair(h, w, seed=1234, dtype=dt)
        flow, st = O.tvl1_calc(I0, I1, O.tvl1_params(**kw), return_stats=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), I0=I0, I1=I1, flow=flow.astype(np.float32),
                            iters=np.array(st["iters"], np.int32), params=np.array(json.dumps(kw)))
        print(name, "EPE vs analytic truth", synth.epe(flow, gt), st["iters"][0])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    tvl1()
