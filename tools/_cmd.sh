timeout 1500 python -m pytest tests/test_baseline_sizes.py tests/test_tvl1_gpu.py -m gpu -x -q 2>&1 | tail -3
