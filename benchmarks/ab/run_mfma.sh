set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 20 --warmup 2"
rm -f gpurun_out/m_*.json
for nq in 4 16 32 64 128; do
$B --nq $nq > gpurun_out/m_f32_nq${nq}_list.json
$B --nq $nq --opt mfma_filter_k=1 > gpurun_out/m_f32_nq${nq}_filt.json
$B --nq $nq --storage bf16 --dim 768 > gpurun_out/m_bf16_nq${nq}_list.json
$B --nq $nq --storage bf16 --dim 768 --opt mfma_filter_k=1 > gpurun_out/m_bf16_nq${nq}_filt.json
done
python benchmarks/show.py gpurun_out/m_*.json | cut -c1-36,96-200
