"""Per-kernel summary and launch timeline of the LAST `marker` kernel onwards in a rocprofv3 --kernel-trace csv.
usage: python tools/trace_summary.py <dir> <marker substring> [timeline file]"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
sel = rows[starts[-1]:]; t0 = int(sel[0]["Start_Timestamp"])
short = lambda n: n.split("(")[0].replace("void ", "").replace("mi::", "")[:44]
busy = 0; prev = 0; gaps = 0
tl = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    busy += e - s; gaps += max(0, s - prev)
    if tl: tl.write(f"{s/1000:9.1f} {(e-s)/1000:8.1f} gap {max(0,s-prev)/1000:7.1f} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {short(r['Kernel_Name'])}\n")
    prev = e
span = max(int(r["End_Timestamp"]) for r in sel) - t0
print("launches", len(sel), "span us", span / 1000, "busy us", busy / 1000, "gaps us", gaps / 1000)
c = collections.Counter(); d = collections.Counter()
for r in sel:
    k = short(r["Kernel_Name"]); c[k] += 1; d[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in d.most_common(14): print(f"{k:46s} n={c[k]:4d} total {v/1000:8.1f} us avg {v/1000/c[k]:7.1f}")
