#!/bin/bash
# Dense stage of the hybrid query (1M x 384, nq 1, k 90): list mode against filter mode and fewer blocks.
set -e
mkdir -p gpurun_out/c5_dense
run() { name=$1; shift
  timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --rows 1000000 --steps 500 --warmup 50 "$@" > gpurun_out/c5_dense/$name.json
  python3 -c "
import json; d=json.load(open('gpurun_out/c5_dense/$name.json')); print('$name ms/step %.4f' % d['ms_per_step'])"
}
run k10 --k 10
run k90 --k 90
run k90_filter --k 90 --opt mfma_filter_k=1
run k90_bpc2 --k 90 --opt stream_bpc=2
run k90_bpc1 --k 90 --opt stream_bpc=1
run k90_rg2 --k 90 --opt stream_rg=2
run k30 --k 30
run k30_filter --k 30 --opt mfma_filter_k=1
