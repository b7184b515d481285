#!/bin/bash
# Round-1 GPU session A: parity tests, bench, kernel trace, PMC passes, time-block sweep.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01a
mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log
(timeout 500 python bench.py 2> $O/bench.err | tail -1) > $O/bench.json
(timeout 300 python tools/sweep_tb.py --batch 16 > $O/sweep.json 2> $O/sweep.err)
BCMD="python bench.py --steps 2 --warmup 1 --no-variants --no-cpu"
(timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace -- $BCMD > $O/ktrace.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --pmc $c -f csv -d $O/pmc_$c -- python bench.py --steps 1 --warmup 0 --no-variants --no-cpu > $O/pmc_$c.log 2>&1)
done
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -f csv -d $O/pmc_SQ -- python bench.py --steps 1 --warmup 0 --no-variants --no-cpu > $O/pmc_SQ.log 2>&1)
(timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -f csv -d $O/pmc_SQ2 -- python bench.py --steps 1 --warmup 0 --no-variants --no-cpu > $O/pmc_SQ2.log 2>&1)
# keep only the small summaries (counter CSVs can be large)
for d in $O/pmc_* ; do [ -d "$d" ] && python tools/pmc_summary.py $d > $d.md 2>/dev/null; done
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -type f -size +4M -delete
ls -la $O
