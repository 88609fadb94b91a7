#!/bin/bash
# A/B of the bf16 tile kernel's MFMA shape on the timing harness (pipe_timing.hip): 32x32x16 (m0) against
# 16x16x32 (m1), C3's contraction on random data, alternating
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
out=gpurun_out/r03/mfma16_ab.txt
: > $out
for rep in 1 2 3; do
  for b in ${@:-m0 m1}; do
    echo "== $b (round $rep)" >> $out
    timeout -k 10 60 ./benchmarks/pipe_timing_$b 1024 768 10000000 4 0 | tail -2 >> $out || exit $?
  done
done
cat $out
