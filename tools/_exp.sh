run() { env "$@" python bench.py --workload farneback --no-cpu --steps 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), d['batched_calc_batch'])"; }
run A=1
run A=2
timeout 300 python -m pytest tests/test_farneback.py tests/test_cpp_shim.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
