set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --steps 30 --warmup 3"
rm -f gpurun_out/s_*.json
for nq in 2 3 4 8; do
$B --nq $nq > gpurun_out/s_scan_nq$nq.json
$B --nq $nq --opt mfma_min_nq=2 > gpurun_out/s_mfma_nq$nq.json
$B --nq $nq --opt mfma_min_nq=2 --opt mfma_shape=1 > gpurun_out/s_mfma1_nq$nq.json
done
$B --nq 2 --storage bf16 --dim 768 > gpurun_out/s_bf_scan_nq2.json
$B --nq 2 --storage bf16 --dim 768 --opt mfma_min_nq=2 > gpurun_out/s_bf_mfma_nq2.json
$B --nq 4 --storage bf16 --dim 768 > gpurun_out/s_bf_scan_nq4.json
$B --nq 4 --storage bf16 --dim 768 --opt mfma_min_nq=2 > gpurun_out/s_bf_mfma_nq4.json
python benchmarks/show.py gpurun_out/s_*.json
