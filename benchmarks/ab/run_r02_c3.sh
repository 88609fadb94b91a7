#!/bin/bash
# C3 with the pipelined kernel: sample pass by its own sample mode (default) against the list-mode tile kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_flat_gpu.py tests/test_flat_bf16_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/c3_test.log 2>&1
rc=$?; tail -5 $O/c3_test.log; [ $rc -ne 0 ] && exit $rc
for ps in 0 -1 0 -1; do
  timeout -k 10 240 python -u bench.py --rows 10000000 --dim 768 --storage bf16 --nq 1024 --k 100 --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-batch-leg --opt mfma_pipe_sample=$ps 2> /dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mfma_pipe_sample=$ps: ms/step %.3f QPS %.0f frac %.4f'%(r['ms_per_step'], r['value'], r['roofline']['frac']))" || exit $?
done
