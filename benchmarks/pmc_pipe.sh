#!/bin/bash
# L2 / fabric counters of the pipelined tile kernel alone (benchmarks/pipe_timing_0); $1 = binary suffix
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B=./benchmarks/pipe_timing_${1:-0}
OUT=gpurun_out/r02/pmc_pipe_${1:-0}
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $OUT/p1 --output-format csv -- $B 1024 768 10000000 4 0 > $OUT/p1.log 2>&1 || exit $?
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/p2 --output-format csv -- $B 1024 768 10000000 4 0 > $OUT/p2.log 2>&1 || exit $?
python3 benchmarks/pmc_summary.py $OUT flat_pipe_kernel > $OUT/summary.json
cat $OUT/summary.json
