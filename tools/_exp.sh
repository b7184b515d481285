timeout 600 python -m pytest tests/test_baseline_sizes.py -m gpu -q -x -p no:cacheprovider -k "iteration_limits" 2>&1 | tail -15
