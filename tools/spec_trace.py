"""Print the decisions of the speculative steps (k_iterate_tbr MODE 1) of a class-default TV-L1 calc: per (scale, warp) and
pair, the sequence of launches as  <block length run>[+accepted by the next launch's settling | R<k> = replay of k].
Usage (GPU box): python tools/spec_trace.py [--pairs 4] [--slack 0] [--calls 2 [--roll 1]]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--slack", type=int, default=0)
    ap.add_argument("--size", default="1080x1920")
    ap.add_argument("--calls", type=int, default=1, help="calcs on the handle before the traced one's slots are read (2: the traced calc has the first one's block-length history)")
    ap.add_argument("--roll", type=int, default=0, help="with --calls > 1: roll the batch by this many pairs between calls (history from other pairs)")
    a = ap.parse_args()
    import torch
    from opencv_contrib_amd import capi, cuda, synth
    h, w = map(int, a.size.split("x"))
    dev = torch.device("cuda:0")
    prs = [synth.flow_pair(h, w, seed=1234 + i) for i in range(a.pairs)]
    I0 = torch.stack([torch.from_numpy(p[0]) for p in prs]).to(dev)
    I1 = torch.stack([torch.from_numpy(p[1]) for p in prs]).to(dev)
    alg = cuda.OpticalFlowDual_TVL1.create(stopSlack=a.slack)
    for c in range(max(a.calls, 1)):
        sh = (a.calls - 1 - c) * a.roll
        alg.calc_batch(torch.roll(I0, sh, 0).contiguous(), torch.roll(I1, sh, 0).contiguous())
    torch.cuda.synchronize()
    cap = 20000
    buf = (C.c_int * (8 * cap))()
    for pair in range(a.pairs):
        n = capi.lib().miflow_selftest_tvl1_slots(alg._h, pair, buf, cap, None)
        if n < 0:
            raise RuntimeError(f"miflow_selftest_tvl1_slots: {n}")
        rows = np.frombuffer(buf, dtype=np.int32, count=8 * n).reshape(n, 8).copy()
        print(f"pair {pair}: {n} launches")
        key = None
        line = []
        for r in rows:
            k = (int(r[0]), int(r[1]))
            if k != key:
                if line:
                    print(f"  s{key[0]} w{key[1]}: " + " ".join(line))
                key, line = k, []
            acc, flip, done = (r[3] >> 8) & 0xff, r[3] & 1, (r[3] >> 2) & 1
            tok = ""
            if acc:
                tok += f"{'R' if flip else '+'}{acc}"
            if r[5]:
                tok += f" run{r[5]}"
            if r[7]:
                tok += f" defer{r[7]}"
            if done and not acc and not r[5]:
                tok += ""
            if tok:
                line.append(f"[{tok.strip()}|n={r[4]}]")
        if line:
            print(f"  s{key[0]} w{key[1]}: " + " ".join(line))
        print("  iterations:", alg.lastIterations(pair))


if __name__ == "__main__":
    main()
