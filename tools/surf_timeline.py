"""Timeline of the LAST detect+describe frame of a `rocprofv3 --kernel-trace` run of `bench.py --workload surf`: every launch with its
start relative to the frame's first kernel, duration, gap to the previous launch's end, grid.  usage: python tools/surf_timeline.py <dir>"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "surf::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a frame starts with k_int_cols; keep the last frame that contains a descriptor kernel
starts = [i for i, r in enumerate(rows) if "k_int_cols" in r["Kernel_Name"]]
frames = [(a, b) for a, b in zip(starts, starts[1:] + [len(rows)]) if any("k_descriptors" in r["Kernel_Name"] for r in rows[a:b])]
a, b = frames[-2] if len(frames) > 1 else frames[-1]
# the fills that precede k_int_cols belong to the frame: walk back over rocclr kernels
while a > 0 and "rocclr" in rows[a - 1]["Kernel_Name"]:
    a -= 1
sel = rows[a:b]
t0 = int(sel[0]["Start_Timestamp"])
prev = t0
tot = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi::surf::", "")[:40]
    print(f"{(s - t0) / 1000:8.1f} {(e - s) / 1000:7.1f} gap {(s - prev) / 1000:6.1f} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']} {name}")
    prev = max(prev, e)
    tot += e - s
print("frame span us:", (prev - t0) / 1000, " sum of kernel times:", tot / 1000)
