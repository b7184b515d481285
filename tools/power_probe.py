"""Socket power and shader clock of the kernels a TV-L1 calc is made of, each held on the chip for a few seconds (rocm-smi polled
beside it): where the energy of a step goes.  MI355X runs this workload AT ITS POWER LIMIT (the clock is what gives), so energy per
pixel -- not issue slots or bytes alone -- is what the headline rate is made of.
Usage: python tools/power_probe.py [seconds per segment]"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from opencv_contrib_amd import capi, cuda, synth

dev = torch.device("cuda", 0)
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


class Poll:
    def __init__(self):
        self.samples, self.stop = [], False
        self.th = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop:
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                p = re.search(r"Power \(W\): ([\d.]+)", o)
                if c and p:
                    self.samples.append((time.perf_counter(), int(c.group(1)), float(p.group(1))))
            except Exception:
                pass
            time.sleep(0.15)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.th.join(timeout=6)

    def mean(self, t0, t1):
        s = [x for x in self.samples if t0 + 0.3 * (t1 - t0) <= x[0] <= t1]   # the settled part
        if not s:
            return None, None, 0
        return sum(x[1] for x in s) / len(s), sum(x[2] for x in s) / len(s), len(s)


def segment(poll, name, fn, work_unit, unit):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SEC:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        n += 4
    t1 = time.perf_counter()
    clk, pw, k = poll.mean(t0, t1)
    rate = n * work_unit / (t1 - t0)
    print(f"{name:58s} {rate:10.3e} {unit}/s   sclk {clk and round(clk)} MHz   power {pw and round(pw)} W   ({k} samples)", flush=True)
    return rate, clk, pw


with Poll() as poll:
    time.sleep(1.5)
    t = time.perf_counter()
    idle = poll.mean(t - 1.5, t)
    print(f"idle: sclk {idle[0]} MHz, power {idle[1]} W")
    a = torch.empty(1 << 30, dtype=torch.float32, device=dev).normal_()     # 4 GiB
    b = torch.empty_like(a)
    segment(poll, "torch copy 4 GiB -> 4 GiB", lambda: b.copy_(a), 2 * a.numel() * 4, "B")
    segment(poll, "torch read-only (sum) 4 GiB", lambda: a.sum(), a.numel() * 4, "B")
    del a, b
    H, W = 4320, 7680
    g = torch.Generator(device="cpu").manual_seed(1)
    r = lambda s=1.0: (torch.rand((H, W), generator=g) * 2 - 1).mul_(s).to(dev)
    ix, iy = r(30.0), r(30.0)
    grad, rc = ix * ix + iy * iy, r(20.0)
    u, p = [r(2.0), r(2.0)], [r(0.7) for _ in range(4)]
    import ctypes as C
    uo, po = [torch.empty_like(x) for x in u], [torch.empty_like(x) for x in p]
    m = capi.mat_from_tensor
    U = (capi.Mat * 2)(*[m(x) for x in u]); P = (capi.Mat * 4)(*[m(x) for x in p])
    UO = (capi.Mat * 2)(*[m(x) for x in uo]); PO = (capi.Mat * 4)(*[m(x) for x in po])

    def it(tb, n):
        capi.check(capi.lib().mi_tvl1_iterate(0, tb, n, C.byref(m(ix)), C.byref(m(iy)), C.byref(m(grad)), C.byref(m(rc)), U, P, UO, PO,
                                              0.045, 0.3, 0.8333, None, capi.current_stream_ptr()))
    segment(poll, f"k_iterate_tbr T=10 (fast math), {W}x{H}, 10 iterations/launch", lambda: it(10, 10), 10.0 * H * W, "px-it")
    segment(poll, f"k_iterate_tbr T=5, 5 iterations/launch", lambda: it(5, 5), 5.0 * H * W, "px-it")
    segment(poll, f"k_iterate (one iteration per launch, fast)", lambda: it(1, 1), 1.0 * H * W, "px-it")
    I0, I1 = r(0.5).add_(0.5), r(0.5).add_(0.5)
    outs = [torch.empty_like(I0) for _ in range(5)]
    wa = [C.byref(m(x)) if x is not None else None for x in (I0, I1, None, None, u[0], u[1], *outs)]
    segment(poll, f"k_warp6 (CPU-class semantics), {W}x{H}", lambda: capi.check(capi.lib().mi_tvl1_warp_backward(0, *wa)), 1.0 * H * W, "px")
    del ix, iy, grad, rc, u, p, uo, po, I0, I1, outs
    B = 64
    pairs = [synth.flow_pair(1080, 1920, seed=50 + i)[:2] for i in range(4)]
    J0 = torch.stack([torch.from_numpy(pairs[i % 4][0]) for i in range(B)]).to(dev)
    J1 = torch.stack([torch.from_numpy(pairs[i % 4][1]) for i in range(B)]).to(dev)
    F = torch.empty((B, 1080, 1920, 2), dtype=torch.float32, device=dev)
    for lanes in (0, 1):
        alg = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0, lanes=lanes)
        segment(poll, f"calc_batch 64 x 1080p, N=10, lanes={lanes or 'auto(2)'}", lambda: alg.calc_batch(J0, J1, F), float(B), "pairs")
        del alg
