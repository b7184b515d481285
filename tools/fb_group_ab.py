"""A/B of MIFLOW_FB_GROUP_MB on the batched Farneback bench (one subprocess per setting) + digests."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mb in ("0", "160", "200", "320", "480", "0", "160"):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "farneback", "--steps", "4", "--warmup", "2", "--no-cpu"],
                       capture_output=True, text=True, env=dict(os.environ, MIFLOW_FB_GROUP_MB=mb), timeout=300)
    try:
        f = json.load(open(os.path.join(ROOT, "bench_full_farneback.json")))
        print("group MB", mb, "batched", round(f["value"], 1), "equals single calcs:", f["batched_calc_batch"]["equals_single_calc"], "sequential", round(f["sequential_calc_pairs_per_s"], 1), flush=True)
    except Exception as e:
        print(mb, "failed", e, r.stderr[-500:])
