#!/bin/bash
# block shape 10: parity test, then the standalone kernel (benchmarks/pipe_timing_*), one binary per line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_flat_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "pipelined" > $O/pp_test.log 2>&1
rc=$?; tail -5 $O/pp_test.log; [ $rc -ne 0 ] && exit $rc
: > $O/pp_small.txt
for b in ${BINS:-0}; do for n in ${NS:-10000000}; do
  echo "== bin $b n $n" >> $O/pp_small.txt
  timeout -k 10 60 ./benchmarks/pipe_timing_$b 1024 768 $n 4 0 | tail -2 >> $O/pp_small.txt || { cat $O/pp_small.txt; exit 1; }
done; done
cat $O/pp_small.txt
