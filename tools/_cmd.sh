timeout 1500 python -m pytest tests/test_baseline_sizes.py tests/test_tvl1_gpu.py tests/test_cpp_shim.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/single_calc_bench.py 2>&1 | tail -4
