set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg"
rm -f gpurun_out/m_*.json
for ns in 2 3; do
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_shape=11 --opt mfma_stages=$ns --opt mfma_filter_k=1 > gpurun_out/m_bf16_k10_s11_filt_ns$ns.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_shape=11 --opt mfma_stages=$ns > gpurun_out/m_bf16_k10_s11_ns$ns.json
$B --storage bf16 --dim 768 --k 100 --nq 1024 --steps 10 --warmup 2 --opt mfma_shape=11 --opt mfma_stages=$ns > gpurun_out/m_bf16_k100_s11_ns$ns.json
$B --nq 1024 --steps 5 --warmup 1 --opt mfma_shape=11 --opt mfma_stages=$ns > gpurun_out/m_f32_k10_s11_ns$ns.json
done
$B --storage bf16 --dim 768 --k 100 --nq 1024 --steps 10 --warmup 2 > gpurun_out/m_bf16_k100_default.json
python benchmarks/show.py gpurun_out/m_*.json | cut -c1-36,96-200
