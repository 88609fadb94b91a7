#!/bin/bash
# Variant builds of the stream kernel (ring depth, block size) x blocks per CU x wave-tile height, nq = 1, 10M x 384.
set -e
mkdir -p gpurun_out/stream_var
run() { lib=$1; bpc=$2; rg=$3
    if [ $lib = default ]; then unset RAG_MI355X_LIB; else export RAG_MI355X_LIB=$PWD/benchmarks/lib_$lib.so; fi
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --steps 200 --warmup 20 --opt stream_bpc=$bpc --opt stream_rg=$rg > gpurun_out/stream_var/${lib}_b${bpc}_rg$rg.json
    python3 -c "
import json; d=json.load(open('gpurun_out/stream_var/${lib}_b${bpc}_rg$rg.json')); print('$lib bpc $bpc rg $rg ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
run nb2 1 4
run nb2 1 2
run nb2 2 2
run default 1 2
for lib in t128nb2 t128nb3 t128nb6; do
  for bpc in 1 2 4; do
    run $lib $bpc 4
  done
  run $lib 2 2
done
