"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-variants --no-cpu
--no-secondary`: mean HBM bytes per launch of the two dominant TV-L1 kernels.

HBM bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024: the counters are in KiB and, on gfx950, FETCH_SIZE reports half the bytes of a
wide coalesced read (MI355X_MICROARCH.md, HBM section); the x 2 is checked below on k_convert, whose byte count is known
(reads 2 x 4 B, writes 2 x 4 B per pixel of the f32 input pair).  Usage: pmc_to_traffic.py <session dir> <source label>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                a = acc[r.get("Kernel_Name", "?")][r.get("Counter_Name", "?")]
                a[0] += float(r.get("Counter_Value", 0) or 0)
                a[1] += 1
    return acc


def main():
    d, label = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64   # pairs per step of the profiled bench command (2 lanes)
    acc = load(d)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}   # keys of other workloads (tools/pmc_secondary.py) are kept

    def mean(kpred, counter):
        s = n = 0
        for k, c in acc.items():
            if kpred(k) and counter in c:
                s += c[counter][0]; n += c[counter][1]
        return (s / n, n) if n else (None, 0)

    # the fixed-work T = 10 kernel: MODE 0 is the sixth template argument, the seventh (JW: 0 / 1 / 2 / 3) the wave arrangement
    import re
    for key, pred in (("tbr", lambda k: re.search(r"k_iterate_tbr<10, 1, (true|false), \d+, \d+, 0(, \d+)?(, (true|false), (true|false))?(, 0)?>\(mi::tvl1::TbArgs\)", k) is not None),
                      ("warp6", lambda k: "k_warp6<" in k), ("convert", lambda k: "k_convert" in k)):
        f, nf = mean(pred, "FETCH_SIZE")
        w, nw = mean(pred, "WRITE_SIZE")
        if f is None or w is None:
            continue
        out[key] = {"hbm_bytes_per_launch": (2 * f + w) * 1024, "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024, "launches": [nf, nw],
                    "pairs_per_launch": batch // 2,
                    "source": f"{label}: mean over the launches of two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of "
                              f"`python bench.py --no-variants --no-cpu --no-secondary --steps 2 --warmup 1` ({batch} pairs, 2 lanes of {batch // 2}): "
                              "(FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE) x 1024"}
    if "convert" in out:   # calibration: 1920 x 1080 x (pairs per lane) x (2 x 4 B read, 2 x 4 B written)
        px = 1920 * 1080 * (batch // 2)
        out["convert"]["expected_fetch_bytes"] = px * 8
        out["convert"]["expected_write_bytes"] = px * 8
    out["v1"] = {"hbm_bytes_per_launch": None, "source": "not collected for the exact-math kernel"}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
