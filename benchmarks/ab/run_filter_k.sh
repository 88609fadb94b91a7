set -e
B="timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 10 --warmup 2"
rm -f gpurun_out/f_*.json
for k in 16 24 32; do for nq in 64 1024; do
$B --nq $nq --k $k --opt mfma_filter_k=1000 > gpurun_out/f_k${k}_nq${nq}_list.json
$B --nq $nq --k $k --opt mfma_filter_k=1 > gpurun_out/f_k${k}_nq${nq}_filt.json
done; done
python benchmarks/show.py gpurun_out/f_*.json | cut -c1-30,96-200
