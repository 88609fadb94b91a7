#!/bin/bash
# Blocks per CU of the stream kernel against batch size and row type (10M rows, k = 10).
set -e
mkdir -p gpurun_out/bpc_nq
for cfg in "f32 384" "bf16 768"; do
 set -- $cfg
 for nq in 1 2 4 8; do
  for bpc in 1 2 3 0; do
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --storage $1 --dim $2 --nq $nq --steps 200 --warmup 20 --opt stream_bpc=$bpc > gpurun_out/bpc_nq/$1_nq${nq}_b$bpc.json
    python3 -c "
import json; d=json.load(open('gpurun_out/bpc_nq/$1_nq${nq}_b$bpc.json')); print('$1 nq $nq bpc $bpc ms/step %.4f' % d['ms_per_step'])"
  done
 done
done
