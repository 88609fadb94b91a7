#!/bin/bash
# The driver's GPU test command, with durations; log under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread --durations=8 > gpurun_out/r02/pytest.log 2>&1
rc=$?
tail -14 gpurun_out/r02/pytest.log
exit $rc
