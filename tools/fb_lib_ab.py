"""A/B of build variants of the library on the Farneback bench (one subprocess per library).  usage: python tools/fb_lib_ab.py libmiflow.so libmiflow__x.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "farneback", "--steps", "4", "--warmup", "2", "--no-cpu"],
                       capture_output=True, text=True, env=dict(os.environ, MIFLOW_LIB=lib), timeout=300)
    try:
        f = json.load(open(os.path.join(ROOT, "bench_full_farneback.json")))
        print(lib, "batched", round(f["value"], 1), "equals single calcs:", f["batched_calc_batch"]["equals_single_calc"], "sequential", round(f["sequential_calc_pairs_per_s"], 1),
              "four streams", round(f["four_streams_pairs_per_s"], 1), flush=True)
    except Exception as e:
        print(lib, "failed", e, r.stderr[-500:])
