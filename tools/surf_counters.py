"""profiles/surf_counters.json from a rocprofv3 --pmc pass (vector-L1 counters) and a --kernel-trace --stats pass of
`python bench.py --workload surf --no-cpu --steps 2`: per-launch means of the SURF kernels, tag lookups per second against one lookup
per clock and CU.  usage: python tools/surf_counters.py <pmc dir> <kernel_stats.csv> <source label>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_descriptors_staged", "k_det_trace_all", "k_nms_flag_all", "k_orientation", "k_poly_build", "k_int_rows_wide", "k_nms_write_all", "k_scan_counts_all")


def main():
    pmc, stats, label = sys.argv[1], sys.argv[2], sys.argv[3]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in glob.glob(os.path.join(pmc, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = next((n for n in KERNELS if n in r["Kernel_Name"]), None)
            if k:
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    dur = {}
    for r in csv.DictReader(open(stats)):
        k = next((n for n in KERNELS if n in r["Name"]), None)
        if k:
            dur[k] = float(r["AverageNs"]) / 1e3
    peak = 256 * 2.4e9
    out = {"source": label, "peak_l1_tag_lookups_per_s": peak,
           "note": "TCP_TOTAL_CACHE_ACCESSES = tag lookups of the per-CU vector L1 (one per 64-B line a wave-level load touches); one lookup per "
                   "clock and CU is the peak (2.4 GHz nominal).", "kernels": {}}
    for k, c in acc.items():
        m = {n: v[0] / max(v[1], 1) for n, v in c.items()}
        look = m.get("TCP_TOTAL_CACHE_ACCESSES_sum")
        e = {"avg_us": dur.get(k), "tcp_total_cache_accesses": look, "tcp_tcc_read_req": m.get("TCP_TCC_READ_REQ_sum"),
             "vmem_read_wave_instructions": m.get("SQ_INSTS_VMEM_RD"), "lds_wave_instructions": m.get("SQ_INSTS_LDS"),
             "tcp_pending_stall_cycles": m.get("TCP_PENDING_STALL_CYCLES_sum")}
        if look and dur.get(k):
            e["l1_tag_lookups_per_s"] = look / (dur[k] * 1e-6)
            e["frac_of_l1_tag_peak"] = e["l1_tag_lookups_per_s"] / peak
        if look and m.get("SQ_INSTS_VMEM_RD"):
            e["lines_per_wave_load"] = look / m["SQ_INSTS_VMEM_RD"]
        if look and m.get("TCP_TCC_READ_REQ_sum") is not None:
            e["l1_hit_rate"] = 1.0 - m["TCP_TCC_READ_REQ_sum"] / look
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(ROOT, "profiles", "surf_counters.json"), "w"), indent=1)
    for k, e in sorted(out["kernels"].items(), key=lambda kv: -(kv[1].get("avg_us") or 0)):
        print(k, {n: (round(v, 3) if isinstance(v, float) else v) for n, v in e.items()})


if __name__ == "__main__":
    main()
