#!/bin/bash
# Ping-pong tile kernel (block shape 10): parity test, standalone timing + ablations, C3 under rocprofv3,
# kernel trace of the default bench (step overhead analysis).
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_flat_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "pipelined" > $O/pp_test.log 2>&1
rc=$?; tail -5 $O/pp_test.log; [ $rc -ne 0 ] && exit $rc
: > $O/pp_timing.txt
for b in 0 1 2 4 v2; do
  echo "== $b" >> $O/pp_timing.txt
  timeout -k 10 60 ./benchmarks/pipe_timing_$b 1024 768 10000000 4 0 | tail -2 >> $O/pp_timing.txt || exit $?
done
cat $O/pp_timing.txt
rm -rf $O/prof_c3_pp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/prof_c3_pp --output-format csv -- python3 bench.py --rows 10000000 --dim 768 --storage bf16 --nq 1024 --k 100 --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-batch-leg --opt mfma_shape=10 > $O/pp_c3.json 2> $O/pp_c3.log || exit $?
cat $O/pp_c3.json
rm -rf $O/prof_default_trace
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/prof_default_trace --output-format csv -- python3 bench.py --no-legs --no-cpu-baseline --no-batch-leg > $O/default_trace.json 2> $O/default_trace.log || exit $?
cat $O/default_trace.json
find $O/prof_default_trace $O/prof_c3_pp -name "*.csv" | head
