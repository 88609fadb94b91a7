set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 20 --warmup 2"
rm -f gpurun_out/w_*.json
for nq in 64 16; do
for sh in 1 2 3 5 6 7; do
for bpc in 1 2 3; do
$B --nq $nq --opt mfma_shape=$sh --opt mfma_bpc=$bpc > gpurun_out/w_nq${nq}_s${sh}_b${bpc}.json || true
done; done; done
python benchmarks/show.py gpurun_out/w_*.json | cut -c1-36,96-200
