#!/bin/bash
# Stream kernel list-mode tail (block merge shared by the four waves): tests + timings.
set -e
mkdir -p gpurun_out/tail_ab
timeout -k 10 900 python3 -u -m pytest tests/test_flat_stream_gpu.py tests/test_flat_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/tail_ab/pytest.log 2>&1 || { tail -30 gpurun_out/tail_ab/pytest.log; exit 1; }
tail -2 gpurun_out/tail_ab/pytest.log
for cfg in "2000 8 10" "2000 1 10" "100000 8 10" "10000000 1 10" "10000000 4 10" "10000000 8 10" "10000000 8 50" "10000000 2 100"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --rows $1 --nq $2 --k $3 --steps 300 --warmup 30 --no-cpu-baseline --no-batch-leg > gpurun_out/tail_ab/r$1_nq$2_k$3.json
  python3 -c "
import json
d=json.load(open('gpurun_out/tail_ab/r$1_nq$2_k$3.json'))
print('rows $1 nq $2 k $3 ms/step %.4f' % d['ms_per_step'], 'kernel_ms', d.get('roofline',{}).get('kernel_ms'))
"
done
