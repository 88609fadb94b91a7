#!/bin/bash
# A/B of two builds of the library on one box, alternating: bash benchmarks/run_lib_ab.sh <other .so> <bench args...>
set -e
other=$1; shift
mkdir -p gpurun_out/lib_ab
for rep in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export RAG_MI355X_LIB=$PWD/$other; else unset RAG_MI355X_LIB; fi
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --steps 300 --warmup 30 "$@" > gpurun_out/lib_ab/${which}_$rep.json
    python3 -c "
import json; d=json.load(open('gpurun_out/lib_ab/${which}_$rep.json')); print('$which $rep ms/step %.4f' % d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms_avg', d['roofline'].get('kernel_ms')))"
  done
done
