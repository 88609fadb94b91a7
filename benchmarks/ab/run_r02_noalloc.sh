#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r02/noalloc.log 2>&1
rc=$?
tail -25 gpurun_out/r02/noalloc.log
exit $rc
