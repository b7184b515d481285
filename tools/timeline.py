"""Timeline of the LAST calc of a `rocprofv3 --kernel-trace` run of bench.py: every kernel launch with its start (relative to
the calc's first launch), duration, queue and grid -- who overlaps whom across the two lanes.
usage: python tools/timeline.py <dir with *kernel_trace.csv> [calc_ms_guess]"""
import csv
import glob
import sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# calcs start with k_convert
starts = [i for i, r in enumerate(rows) if "k_convert" in r["Kernel_Name"]]
# two lanes => two k_convert per calc; group converts closer than 1 ms
groups = []
for i in starts:
    t = int(rows[i]["Start_Timestamp"])
    if groups and t - groups[-1][1] < 2_000_000:
        groups[-1][1] = t
    else:
        groups.append([i, t])
first = groups[-1][0]
sel = rows[first:]
t0 = int(sel[0]["Start_Timestamp"])
short = lambda n: n.split("(")[0].replace("void mi::tvl1::", "").replace("mi::tvl1::", "")[:34]
qkey = "Queue_Id" if "Queue_Id" in sel[0] else None
print("cols:", ",".join(sel[0].keys()))
busy_end = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:8.1f} q={r.get(qkey,'?'):>3} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg={r.get('Workgroup_Size_X','?')} {short(r['Kernel_Name'])}")
print("calc span us:", (max(int(r["End_Timestamp"]) for r in sel) - t0) / 1000)
