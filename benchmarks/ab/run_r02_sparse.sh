#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_sparse_gpu.py tests/test_hybrid_full_gpu.py tests/test_host_logic.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r02/sparse_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r02/sparse_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python benchmarks/sparse_ab.py 2> gpurun_out/r02/sparse_ab.err | tee gpurun_out/r02/sparse_ab.txt
