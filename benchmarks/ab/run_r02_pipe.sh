#!/bin/bash
# Block shape 10 (software-pipelined bf16 tile kernel): parity test, then C3 A/B against shape 4.
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -u -m pytest tests/test_flat_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "pipelined" > gpurun_out/r02/pipe_test.log 2>&1
rc=$?
tail -5 gpurun_out/r02/pipe_test.log
[ $rc -ne 0 ] && exit $rc
for shape in 4 10; do
  timeout -k 10 240 python -u bench.py --rows 10000000 --dim 768 --storage bf16 --nq 1024 --k 100 --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-batch-leg --opt mfma_shape=$shape > gpurun_out/r02/pipe_c3_shape$shape.json 2> gpurun_out/r02/pipe_c3_shape$shape.log || exit $?
  python - <<PY
import json
r = json.loads(open("gpurun_out/r02/pipe_c3_shape$shape.json").read().strip().splitlines()[-1])
print("shape $shape: ms/step %.3f  roofline %s" % (r["ms_per_step"], {k: r["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "launches")}))
PY
done
