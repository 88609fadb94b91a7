set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg"
$B --storage bf16 --dim 768 --k 100 --nq 1024 --steps 10 --warmup 2 > gpurun_out/c3_nq1024.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 > gpurun_out/c3_nq1024_k10.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_shape=8 > gpurun_out/c3_nq1024_k10_s8.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_shape=3 > gpurun_out/c3_nq1024_k10_s3.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_stages=2 > gpurun_out/c3_nq1024_k10_ns2.json
$B --storage bf16 --dim 768 --k 10 --nq 64 --steps 20 --warmup 2 > gpurun_out/c3_nq64_k10.json
$B --nq 1024 --steps 5 --warmup 1 > gpurun_out/c_f32_nq1024.json
$B --nq 64 --steps 20 --warmup 2 > gpurun_out/c_f32_nq64.json
$B --nq 16 --steps 20 --warmup 2 > gpurun_out/c_f32_nq16.json
