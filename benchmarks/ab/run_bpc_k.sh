#!/bin/bash
# Blocks per CU of the stream kernel against k and shard size (nq = 1).
set -e
mkdir -p gpurun_out/bpc_k
for rows in 1000000 10000000; do
 for k in 10 30 90 128; do
  for bpc in 0 2 3; do
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --rows $rows --k $k --steps 300 --warmup 30 --opt stream_bpc=$bpc > gpurun_out/bpc_k/r${rows}_k${k}_b$bpc.json
    python3 -c "
import json; d=json.load(open('gpurun_out/bpc_k/r${rows}_k${k}_b$bpc.json')); print('rows $rows k $k bpc $bpc ms/step %.4f' % d['ms_per_step'])"
  done
 done
done
