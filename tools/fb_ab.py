"""A/B of the Farneback launch-form switches on the single-pair / batched bench (one subprocess per setting)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for cfg in ({}, {"MIFLOW_FB_PAIR": "1"}, {"MIFLOW_FB_PAIR": "0"}, {"MIFLOW_FB_NARROW": "1"}, {"MIFLOW_FB_NARROW": "0"}, {"MIFLOW_FB_PAIR": "1", "MIFLOW_FB_NARROW": "1"}):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "farneback", "--steps", "4", "--warmup", "2", "--no-cpu"],
                       capture_output=True, text=True, env=dict(os.environ, **cfg), timeout=300)
    try:
        f = json.load(open(os.path.join(ROOT, "bench_full_farneback.json")))
        print(cfg, "batched", round(f["value"], 1), "sequential", round(f["sequential_calc_pairs_per_s"], 1), "four streams", round(f["four_streams_pairs_per_s"], 1), flush=True)
    except Exception as e:
        print(cfg, "failed", e, r.stderr[-500:])
