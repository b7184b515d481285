"""Digest of what the blocked iteration kernel produces on a list of shapes (stage-level entry mi_tvl1_iterate, 10 fused iterations per
launch, two launches) and of whole calcs: run once per value of MIFLOW_TB_JW (0 = independent waves, 1 = tags, 2 = barrier intervals,
3 = eight joined waves, 4 = barrier intervals with branch-free publishes and mask-free interior blocks; the switch is read once per
process) and compare the outputs -- every joined-wave kernel must be bit-identical to the independent-wave kernel.
Usage: MIFLOW_TB_JW=<0|1|2|3|4> python tools/jw_check.py > digest.txt"""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from opencv_contrib_amd import capi, cuda, synth

dev = torch.device("cuda", 0)


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    return h.hexdigest()[:16]


def planes(h, w, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda s=1.0: (torch.rand((h, w), generator=g) * 2 - 1).mul_(s).to(dev)
    ix, iy = r(30.0), r(30.0)
    grad = ix * ix + iy * iy
    rc = r(20.0)
    u = [r(2.0), r(2.0)]
    p = [r(0.7), r(0.7), r(0.7), r(0.7)]
    return ix, iy, grad, rc, u, p


QUICK = "--quick" in sys.argv   # the shapes only (tests/test_tvl1_gpu.py runs this form in two subprocesses)
shapes = [(40, 64), (37, 65), (90, 246), (90, 247), (64, 256), (33, 257), (50, 300), (120, 482), (77, 483), (31, 500), (200, 720),
          (45, 1000), (270, 1920), (1080, 1920), (442, 786), (16, 16), (21, 1229), (300, 4000)]
if QUICK:
    shapes = [s_ for s_ in shapes if s_[0] * s_[1] <= 600000]
for (h, w) in shapes:
    ix, iy, grad, rc, u, p = planes(h, w, h * 10007 + w)
    for niter in (10, 20):
        uo, po, _ = cuda.tvl1_iterate(ix, iy, grad, rc, u, p, 0.045, 0.3, 0.8333, niter=niter, exact=False, time_block=10, want_err=False)
        print(f"iterate {w}x{h} n={niter} u {digest(uo)} p {digest(po)}", flush=True)
# the speculative steps of the convergence-checked path (MODE 1) are joined under MIFLOW_TB_JW=2 as well: class defaults on pairs whose
# levels span one to three 256-column strips, in one batch
sp = [synth.flow_pair(h_, w_, seed=77 + i)[:2] for i, (h_, w_) in enumerate([(300, 531), (300, 531), (300, 531), (300, 531)])]
algd = cuda.OpticalFlowDual_TVL1.create()
fd = algd.calc_batch([torch.from_numpy(a).to(dev) for a, _ in sp], [torch.from_numpy(b).to(dev) for _, b in sp])
torch.cuda.synchronize()
print("iterate-spec calc 531x300 x4 class defaults", digest([fd]), flush=True)
f1 = cuda.OpticalFlowDual_TVL1.create(iterations=40, epsilon=0.02).calc(torch.from_numpy(sp[0][0]).to(dev), torch.from_numpy(sp[0][1]).to(dev))
torch.cuda.synchronize()
print("iterate-spec calc 531x300 N=40 eps=0.02", digest([f1]), flush=True)
if QUICK:
    fault = C.c_int(-1)
    capi.check(capi.lib().miflow_selftest_jw_fault(C.byref(fault)))
    print("jw_fault", fault.value, flush=True)
    sys.exit(0)
# whole calcs: 1080p batch (both lanes, all five levels), a 4K pair, class defaults on a small pair (MODE 1 is not joined: unchanged)
pairs = [synth.flow_pair(1080, 1920, seed=1234 + i)[:2] for i in range(3)]
I0 = [torch.from_numpy(pairs[i % 3][0]).to(dev) for i in range(6)]
I1 = [torch.from_numpy(pairs[i % 3][1]).to(dev) for i in range(6)]
alg = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0)
f = alg.calc_batch(I0, I1)
torch.cuda.synchronize()
print("calc 1080p x6 N=10", digest([f]), flush=True)
alg30 = cuda.OpticalFlowDual_TVL1.create(iterations=30, epsilon=0.0)
f = alg30.calc_batch(I0[:4], I1[:4])
torch.cuda.synchronize()
print("calc 1080p x4 N=30", digest([f]), flush=True)
A0, A1, _ = synth.flow_pair(2160, 3840, seed=7, flow_scale=3.0, sigma=6.0)
f = alg.calc(torch.from_numpy(A0).to(dev), torch.from_numpy(A1).to(dev))
torch.cuda.synchronize()
print("calc 4K N=10", digest([f]), flush=True)
t0 = time.perf_counter()
for _ in range(3):
    alg.calc_batch(I0, I1)
torch.cuda.synchronize()
print(f"# 3 x calc_batch(6): {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
fault = C.c_int(-1)
capi.check(capi.lib().miflow_selftest_jw_fault(C.byref(fault)))
print("jw_fault", fault.value, flush=True)
