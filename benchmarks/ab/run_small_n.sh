#!/bin/bash
# Small corpora (BASELINE config 0: the reference's own deployment size): latency of one search call.
set -e
mkdir -p gpurun_out/small_n
for rows in 2000 20000 100000 300000; do
  for nq in 1 8; do
    timeout -k 10 120 python3 bench.py --rows $rows --nq $nq --steps 2000 --warmup 100 --no-cpu-baseline --no-batch-leg \
      > gpurun_out/small_n/r${rows}_nq${nq}.json
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/small_n/r${rows}_nq${nq}.json'))
print('rows', $rows, 'nq', $nq, 'ms/step %.4f' % d['ms_per_step'], 'kernel', d.get('roofline',{}).get('kernel'), 'kernel_ms', d.get('roofline',{}).get('kernel_ms'))
"
  done
done
