#!/bin/bash
# NB = 2 build of the stream kernel against the default build over batch sizes, row types, k and shard sizes.
set -e
mkdir -p gpurun_out/nb2
run() { lib=$1; bpc=$2; shift 2; tag=$(echo "$@" | tr -d ' -')
    if [ $lib = default ]; then unset RAG_MI355X_LIB; else export RAG_MI355X_LIB=$PWD/benchmarks/lib_$lib.so; fi
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --steps 200 --warmup 20 --opt stream_bpc=$bpc "$@" > gpurun_out/nb2/${lib}_b${bpc}_$tag.json
    python3 -c "
import json; d=json.load(open('gpurun_out/nb2/${lib}_b${bpc}_$tag.json')); print('$lib bpc $bpc [$*] ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
for args in "--nq 2" "--nq 4" "--nq 8" "--k 100" "--nq 4 --k 50" "--storage bf16 --dim 768" "--storage bf16 --dim 768 --nq 4" "--storage bf16 --dim 768 --nq 8" "--storage bf16 --dim 768 --k 100" "--dim 768 --rows 5000000" "--dim 100 --rows 20000000" "--rows 1000000 --nq 8" "--rows 100000"; do
  run default 0 $args
  run nb2 1 $args
  run nb2 2 $args
done
