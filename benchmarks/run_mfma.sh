set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg"
rm -f gpurun_out/m_*.json
$B --storage bf16 --dim 768 --k 100 --nq 1024 --steps 10 --warmup 2 $EXTRA > gpurun_out/m_c3_k100.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 $EXTRA > gpurun_out/m_c3_k10.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_shape=8 $EXTRA > gpurun_out/m_c3_k10_s8.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_filter_k=1 $EXTRA > gpurun_out/m_c3_k10_filt.json
$B --storage bf16 --dim 768 --k 10 --nq 1024 --steps 10 --warmup 2 --opt mfma_filter_k=1 --opt mfma_shape=8 $EXTRA > gpurun_out/m_c3_k10_filt_s8.json
$B --nq 1024 --steps 5 --warmup 1 $EXTRA > gpurun_out/m_f32_nq1024.json
$B --nq 1024 --steps 5 --warmup 1 --opt mfma_filter_k=1 $EXTRA > gpurun_out/m_f32_nq1024_filt.json
$B --nq 64 --steps 20 --warmup 2 $EXTRA > gpurun_out/m_f32_nq64.json
python benchmarks/show.py gpurun_out/m_*.json
