#!/bin/bash
# builds benchmarks/pipe_timing_<e> (ablations RAG_PIPE_EXP, see pipe_timing.hip) and
# benchmarks/pipe_timing_v<n> (schedule variants RAG_PIPE_VAR)
set -e
cd "$(dirname "$0")/.."
mkdir -p build
# (ablations with bit 4 -- no MFMA -- are NOT built by default: with the 16x16x32 path of r03 the run of
# pipe_timing_4 ended in "an illegal memory access" and cost a GPU box; cause not found, do not run it on shared hardware)
for e in ${EXPS:-0 1 2 8 3}; do
  ( f=""; [ $e -ne 0 ] && f="-DRAG_PIPE_EXP=$e"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $f -Iinclude -o benchmarks/pipe_timing_$e benchmarks/pipe_timing.hip rag_system_amd/csrc/lib.hip 2> build/pipe_timing_$e.log || { tail -20 build/pipe_timing_$e.log; exit 1; } ) &
done
for v in ${VARS:-}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRAG_PIPE_VAR=$v -Iinclude -o benchmarks/pipe_timing_v$v benchmarks/pipe_timing.hip rag_system_amd/csrc/lib.hip 2> build/pipe_timing_v$v.log || { tail -20 build/pipe_timing_v$v.log; exit 1; } ) &
done
wait
ls -la benchmarks/pipe_timing_*
