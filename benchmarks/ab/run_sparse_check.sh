#!/bin/bash
# Sparse / hybrid check: GPU tests of the sparse entry points and the host mirror, then the C5 hybrid bench.
set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_sparse_gpu.py tests/test_robustness_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/sparse_check_pytest.log 2>&1 || { tail -30 gpurun_out/sparse_check_pytest.log; exit 1; }
tail -2 gpurun_out/sparse_check_pytest.log
timeout -k 10 300 python3 benchmarks/bench_hybrid.py --no-cpu > gpurun_out/hyb_1m_b.json 2> gpurun_out/hyb_1m_b.log || { tail -20 gpurun_out/hyb_1m_b.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/hyb_1m_b.json')); print('hybrid', round(j['value'],1), 'QPS', {k:round(v,3) for k,v in j['stage_ms'].items()}, 'parity queries', j['parity']['queries_checked'])"
