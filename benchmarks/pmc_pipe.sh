#!/bin/bash
# LDS / wait / issue counters of the standalone tile kernel; $1 = suffix of the timing binary
# (benchmarks/pipe_timing_<suffix>), $2 = output directory under gpurun_out/
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B=./benchmarks/pipe_timing_${1:-0}
OUT=gpurun_out/${2:-r03/pmc_pipe_${1:-0}}
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/p1 --output-format csv -- $B 1024 768 10000000 4 0 > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MFMA SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/p2 --output-format csv -- $B 1024 768 10000000 4 0 > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max -d $OUT/p3 --output-format csv -- $B 1024 768 10000000 4 0 > $OUT/p3.log 2>&1 || { tail -5 $OUT/p3.log; exit 1; }
python3 benchmarks/pmc_summary.py $OUT flat_ > $OUT/summary.json
cat $OUT/summary.json
