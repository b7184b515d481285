timeout 600 python -m pytest tests/test_baseline_sizes.py -m gpu -q -x -p no:cacheprovider -k "lanes or multi_device or batch_of_64" 2>&1 | tail -3
run() { python bench.py --no-variants --no-cpu --no-secondary --steps 8 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"; }
run --lanes 2
run --lanes 3
run --lanes 4
run --lanes 2
run --lanes 3 --batch 24
run --lanes 2 --batch 24
run --defaults --lanes 2
run --defaults --lanes 3
run --defaults --lanes 4
