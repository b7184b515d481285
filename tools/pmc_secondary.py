"""Adds the secondary workloads' dominant kernels to profiles/pmc_traffic.json from separate rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) of `bench.py --workload {stereobm,farneback,surf} --no-cpu`: mean HBM bytes per launch,
(FETCH_SIZE x 2 [gfx950 correction, see pmc_to_traffic.py] + WRITE_SIZE) x 1024.  For Farneback the launches of the finest level
of the batched calc (the largest grid) are taken.  Usage: pmc_secondary.py <session dir> <source label>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"stereobm": ("stereobm", lambda k: "k_block_match" in k),
        "farneback": ("farneback_iterate_level0_batch", lambda k: "k_iterate_t<" in k),
        "surf": ("surf_det_trace", lambda k: "k_det_trace" in k)}


def main():
    d, label = sys.argv[1], sys.argv[2]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for wl, (key, pred) in KEYS.items():
        vals = {}
        split = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = []
            for p in glob.glob(os.path.join(d, f"pmc_{wl}_{counter}", "**", "*counter_collection.csv"), recursive=True):
                with open(p, newline="") as f:
                    for r in csv.DictReader(f):
                        if r.get("Counter_Name") == counter and pred(r.get("Kernel_Name", "")):
                            g = int(r.get("Grid_Size", 0) or 0)
                            rows.append((g, float(r.get("Counter_Value", 0) or 0)))
            if not rows:
                continue
            gmax = max(g for g, _ in rows)
            if wl == "farneback":
                # the finest level of the 640 x 480 batch: 3 x 120 workgroups of 256 per pair; since round 5 a launch covers a pair GROUP
                # (the coarser level with all 32 pairs has the larger grid), so: the most frequent such grid
                fine = [(g, v) for g, v in rows if g % 92160 == 0 and g // 92160 <= 32]
                cnt = defaultdict(int)
                for g, _ in fine:
                    cnt[g] += 1
                gsel = max(cnt, key=cnt.get) if cnt else gmax
                fb_pairs = gsel // 92160 if cnt else None
                sel = [v for g, v in rows if g == gsel]
            else:
                sel = [v for _, v in rows]
            vals[counter] = (sum(sel) / len(sel), len(sel))
            if wl == "stereobm":
                # the mean over ALL launches mixes compute() (one pair per launch) with compute_batch() (B pairs per
                # launch, the largest grid); keep the two apart
                gmin = min(g for g, _ in rows)
                one = [v for g, v in rows if g == gmin]
                many = [v for g, v in rows if g == gmax]
                split[counter] = {"one_pair_launch": (sum(one) / len(one), len(one), gmin), "batched_launch": (sum(many) / len(many), len(many), gmax)}
        if len(vals) == 2:
            f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
            out[key] = {"hbm_bytes_per_launch": (2 * f + w) * 1024, "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024,
                        "launches": [vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]],
                        "source": f"{label}: two separate rocprofv3 --pmc passes of `python bench.py --workload {wl} --no-cpu --steps 2 --warmup 1`, "
                                  "(FETCH_SIZE x 2 + WRITE_SIZE) x 1024, mean over the launches"}
            if wl == "farneback" and fb_pairs:
                out[key]["pairs_per_launch"] = fb_pairs
            if len(split) == 2:
                for kind in ("one_pair_launch", "batched_launch"):
                    ff, nf, g = split["FETCH_SIZE"][kind]
                    ww, nw, _ = split["WRITE_SIZE"][kind]
                    out[key][kind] = {"fetch_bytes": 2 * ff * 1024, "write_bytes": ww * 1024, "hbm_bytes": (2 * ff + ww) * 1024, "launches": [nf, nw], "grid_size": g}
                b = int(os.environ.get("PMC_SBM_BATCH", "0"))
                if b > 0:
                    out[key]["batched_launch"]["pairs"] = b
                    out[key]["batched_launch"]["hbm_bytes_per_pair"] = out[key]["batched_launch"]["hbm_bytes"] / b
            print(key, out[key])
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
