"""Mean of the PMC counters of one kernel from rocprofv3 --pmc csv output(s).  usage: python tools/pmc_kernel.py <dir> <kernel substring> [grid size = total work-items]"""
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"] and (len(sys.argv) < 4 or r["Grid_Size"] == sys.argv[3]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:28s} n={len(v):5d} mean {sum(v)/len(v):16.1f}")
