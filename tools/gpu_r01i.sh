#!/bin/bash
# Round-1 GPU session I: LDS-tile warp kernel A/B + tests.
set -u
export TMPDIR=/tmp
O=gpurun_out/r01i
mkdir -p $O
(timeout 900 python -m pytest tests/test_tvl1_gpu.py tests/test_golden.py tests/test_cpp_shim.py -m gpu -x -q 2>&1 | tail -15) > $O/pytest_tvl1.log
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_warplds.json
(MIFLOW_WARP=1 timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_warp1.json
(timeout 300 python bench.py --no-variants --no-cpu 2>/dev/null | tail -1) > $O/bench_warplds_b.json
BCMD="python bench.py --steps 2 --warmup 1 --no-variants --no-cpu"
(timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/ktrace -- $BCMD > $O/ktrace.log 2>&1)
find $O/ktrace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -f csv -d $O/pmc_SQ -- python bench.py --steps 1 --warmup 0 --no-variants --no-cpu > $O/pmc_SQ.log 2>&1)
python tools/pmc_summary.py $O/pmc_SQ > $O/pmc_SQ.md 2>/dev/null
(timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -f csv -d $O/pmc_SQ2 -- python bench.py --steps 1 --warmup 0 --no-variants --no-cpu > $O/pmc_SQ2.log 2>&1)
python tools/pmc_summary.py $O/pmc_SQ2 > $O/pmc_SQ2.md 2>/dev/null
find $O -type f -size +4M -delete
ls -la $O
