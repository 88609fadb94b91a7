#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02/pipe_vars.txt
: > $out
for b in $@; do
  echo "== $b" >> $out
  timeout -k 10 60 ./benchmarks/pipe_timing_$b 1024 768 10000000 4 0 | tail -2 >> $out || exit $?
done
cat $out
