#!/bin/bash
# Round-1 GPU session M: XCD-aware strip mapping in the temporally blocked kernel -- parity, sweep, traffic counters.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01m
mkdir -p $O
(timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_golden.py -m gpu -q 2>&1 | tail -5) > $O/pytest_tvl1.log
(timeout 300 python tools/sweep_tb.py --no-v1 --tag defaults 2>/dev/null | tail -1) > $O/sweep.jsonl
for v in "2,3,1:5" "2,2,2:5" "1,4,1:8" "1,3,2:8,10" "2,1,1:8,10"; do
  var=${v%%:*}; blocks=${v##*:}
  (MIFLOW_TB_VARIANT=$var timeout 200 python tools/sweep_tb.py --blocks $blocks --tag "variant=$var" --no-v1 2>/dev/null | tail -1) >> $O/sweep.jsonl
done
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench.json
R=$PWD
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/$O/pmc_fetch -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $R/$O/pmc_write -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.md 2>$O/pmc_summary.err
find $O -type f -size +4M -delete
cat $O/pytest_tvl1.log; cat $O/sweep.jsonl; cat $O/bench.json; head -8 $O/pmc_summary.md
