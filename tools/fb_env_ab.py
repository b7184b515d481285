"""A/B of environment settings on the Farneback bench (one subprocess per setting).  usage: python tools/fb_env_ab.py [lib] "A=1 B=2" "A=3" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1]
for spec in sys.argv[2:]:
    env = dict(os.environ, MIFLOW_LIB=lib)
    env.update(dict(kv.split("=", 1) for kv in spec.split()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "farneback", "--steps", "4", "--warmup", "2", "--no-cpu"],
                       capture_output=True, text=True, env=env, timeout=300)
    try:
        f = json.load(open(os.path.join(ROOT, "bench_full_farneback.json")))
        print(f"{spec:50s} batched", round(f["value"], 1), "equals single calcs:", f["batched_calc_batch"]["equals_single_calc"], "sequential", round(f["sequential_calc_pairs_per_s"], 1), flush=True)
    except Exception as e:
        print(spec, "failed", e, r.stderr[-500:])
