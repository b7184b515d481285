"""Aggregate rocprofv3 --pmc CSV output (…_counter_collection.csv) per kernel:
mean counter value per dispatch and dispatch count.  Usage: pmc_summary.py <dir-or-csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def files(arg):
    if os.path.isdir(arg):
        return sorted(glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True))
    return [arg]


def summarize(paths):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in paths:
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                k = r.get("Kernel_Name", "?")
                c = r.get("Counter_Name", "?")
                v = float(r.get("Counter_Value", 0) or 0)
                a = acc[k][c]
                a[0] += v
                a[1] += 1
    out = ["| kernel | counter | dispatches | mean per dispatch | total |", "|---|---|---:|---:|---:|"]
    for k in sorted(acc, key=lambda k: -sum(v[0] for v in acc[k].values())):
        for c, (s, n) in sorted(acc[k].items()):
            out.append(f"| `{k[:90]}` | {c} | {n} | {s / max(n, 1):.6g} | {s:.6g} |")
    return "\n".join(out)


if __name__ == "__main__":
    ps = []
    for a in sys.argv[1:]:
        ps += files(a)
    print(summarize(ps))
