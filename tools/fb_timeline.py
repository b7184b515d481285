"""Per-kernel totals of the LAST batched Farneback calc of a `rocprofv3 --kernel-trace` run (starts at its last k_convert_batch burst).
usage: python tools/fb_timeline.py <dir with *kernel_trace.csv> [full]"""
import csv, glob, sys
from collections import OrderedDict
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
conv = [i for i, r in enumerate(rows) if "k_convert_batch" in r["Kernel_Name"]]
first = conv[-1]
while first - 1 in conv:
    first -= 1
sel = rows[first:]
t0 = int(sel[0]["Start_Timestamp"])
short = lambda n: n.split("(")[0].replace("void mi::fb::", "").replace("mi::fb::", "").replace("void mi::tvl1::", "")[:40]
agg = OrderedDict()
for r in sel:
    k = (short(r["Kernel_Name"]), f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    if len(sys.argv) > 2:
        print(f"{(int(r['Start_Timestamp'])-t0)/1000:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:8.1f} {k[1]:>16} {k[0]}")
span = (max(int(r["End_Timestamp"]) for r in sel) - t0) / 1000
busy = sum(a[1] for a in agg.values())
for k, a in agg.items():
    print(f"{a[0]:4d} x {a[1]/a[0]:8.1f} us = {a[1]:8.1f} us  {k[1]:>16}  {k[0]}")
print(f"span {span:.1f} us, kernels {busy:.1f} us, gaps {span-busy:.1f} us, launches {len(sel)}")
