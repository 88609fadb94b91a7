#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02/pipe_timing.txt
: > $out
for e in 0 1 2 4 8 3 5 6; do
  timeout -k 10 60 ./benchmarks/pipe_timing_$e 1024 768 10000000 4 0 | tail -2 >> $out || exit $?
done
timeout -k 10 60 ./benchmarks/pipe_timing_0 1024 768 10000000 3 0 | tail -1 >> $out || exit $?
timeout -k 10 60 ./benchmarks/pipe_timing_0 1024 768 10000000 4 1 | tail -1 >> $out || exit $?
timeout -k 10 60 ./benchmarks/pipe_timing_0 256 768 10000000 4 0 | tail -1 >> $out || exit $?
cat $out
