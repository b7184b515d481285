timeout 900 python -m pytest tests/test_baseline_sizes.py tests/test_tvl1_gpu.py -m gpu -x -q -k "previous_calc or host_feedback or speculative or tile or convergence" 2>&1 | tail -3
timeout 300 python tools/single_calc_bench.py 2>&1 | tail -4
timeout 200 python tools/tvl1_single.py 1920 1080 20 | cut -c1-90;  timeout 200 python tools/tvl1_single.py 640 480 20 | cut -c1-90
