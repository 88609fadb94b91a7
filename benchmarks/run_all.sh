# tests + the bench lines tracked between kernel changes
set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { tail -30 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
rm -f gpurun_out/c*.json gpurun_out/b_*.json
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg"
$B > gpurun_out/b_nq1.json
$B --k 100 > gpurun_out/b_nq1_k100.json
$B --nq 4 > gpurun_out/b_nq4.json
$B --nq 16 --steps 20 --warmup 2 > gpurun_out/b_nq16.json
$B --nq 64 --steps 20 --warmup 2 > gpurun_out/b_nq64.json
$B --nq 1024 --steps 5 --warmup 1 > gpurun_out/b_nq1024.json
$B --rows 1000000 --steps 200 > gpurun_out/c2_nq1.json
$B --rows 1000000 --nq 64 --steps 50 --warmup 5 > gpurun_out/c2_nq64.json
$B --storage bf16 --dim 768 --k 100 --nq 1 > gpurun_out/c3_nq1.json
$B --storage bf16 --dim 768 --k 100 --nq 1024 --steps 10 --warmup 2 > gpurun_out/c3_nq1024.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 > gpurun_out/c3_nq1024_k10.json
python benchmarks/show.py gpurun_out/b_*.json gpurun_out/c*.json
timeout -k 10 300 python benchmarks/bench_hybrid.py --no-cpu > gpurun_out/hyb_1m_b.json 2> gpurun_out/hyb_1m_b.log
python -c "
import json; j=json.load(open('gpurun_out/hyb_1m_b.json')); print('hybrid', round(j['value'],1), 'QPS', {k:round(v,3) for k,v in j['stage_ms'].items()}, 'parity queries', j['parity']['queries_checked'])"
