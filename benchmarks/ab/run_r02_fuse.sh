#!/bin/bash
# fused merge of the stream kernel: tests, then the default bench with and without it
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_flat_stream_gpu.py tests/test_flat_gpu.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/fuse_test.log 2>&1
rc=$?; tail -8 $O/fuse_test.log; [ $rc -ne 0 ] && exit $rc
for f in 0 -1 0 -1; do
  timeout -k 10 240 python -u bench.py --no-legs --no-cpu-baseline --no-batch-leg --opt stream_fuse=$f 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream_fuse=$f', 'QPS %.2f ms/step %.4f kernel %.4f'%(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms']))" || exit $?
done
for f in 0 -1; do
  timeout -k 10 240 python -u bench.py --rows 1000000 --k 90 --steps 400 --no-legs --no-cpu-baseline --no-batch-leg --opt stream_fuse=$f 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1M k=90 stream_fuse=$f', 'QPS %.2f ms/step %.4f kernel %.4f'%(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms']))" || exit $?
done
