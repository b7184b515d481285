#!/bin/bash
# Round-1 GPU session N: strip->XCD mapping modes of the temporally blocked kernel (MIFLOW_TB_SWZ = 0 | 1 | 2).
set -u
export TMPDIR=/tmp
O=gpurun_out/r01n
mkdir -p $O
for z in 0 2 1; do
  (MIFLOW_TB_SWZ=$z timeout 300 python tools/sweep_tb.py --blocks 5,8,10 --no-v1 --tag swz=$z 2>/dev/null | tail -1) >> $O/sweep.jsonl
done
R=$PWD
cd /tmp
for z in 0 2; do
MIFLOW_TB_SWZ=$z timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/$O/pmc_fetch$z -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
MIFLOW_TB_SWZ=$z timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $R/$O/pmc_write$z -- python $R/bench.py --no-variants --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
done
cd $R
for z in 0 2; do python tools/pmc_summary.py $O/pmc_fetch$z $O/pmc_write$z | head -4 > $O/pmc_summary$z.md; done
find $O -type f -size +4M -delete
cat $O/sweep.jsonl; cat $O/pmc_summary*.md
