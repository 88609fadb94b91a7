#!/bin/bash
# C3 (and its fp32 cousin) with block shape 4 against block shape 10; then the whole GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
for cfg in "bf16 768 4" "bf16 768 10" "f32 384 4" "f32 384 10"; do
  set -- $cfg
  timeout -k 10 240 python -u bench.py --rows 10000000 --dim $2 --storage $1 --nq 1024 --k 100 --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-batch-leg --opt mfma_shape=$3 > $O/c3_$1_shape$3.json 2> $O/c3_$1_shape$3.log || exit $?
  python - <<PY
import json
r = json.loads(open("$O/c3_$1_shape$3.json").read().strip().splitlines()[-1])
print("$1 shape $3: ms/step %.3f  roofline %s" % (r["ms_per_step"], {k: r["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "launches")}))
PY
done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $O/pytest_all.log 2>&1
rc=$?; tail -12 $O/pytest_all.log; exit $rc
