#!/bin/bash
# One and two queries through the stream kernel (mfma_min_nq=1) against the scan kernel.
mkdir -p gpurun_out/nq1s
run() { local name=$1; shift
    timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 50 --warmup 5 "$@" \
        > gpurun_out/nq1s/$name.json 2> gpurun_out/nq1s/$name.err || { echo "$name failed"; tail -3 gpurun_out/nq1s/$name.err; }; }
for k in 10 100; do
  for nq in 1 2; do
    run scan_f32_nq${nq}_k$k --nq $nq --k $k --opt mfma_min_nq=3
    run st_f32_nq${nq}_k$k --nq $nq --k $k --opt mfma_min_nq=1
    run st_f32_nq${nq}_k${k}_list --nq $nq --k $k --opt mfma_min_nq=1 --opt mfma_filter_k=1000
    run scan_bf16_nq${nq}_k$k --nq $nq --k $k --storage bf16 --dim 768 --opt mfma_min_nq=3
    run st_bf16_nq${nq}_k$k --nq $nq --k $k --storage bf16 --dim 768 --opt mfma_min_nq=1
  done
done
run scan_c2 --rows 1000000 --steps 200 --opt mfma_min_nq=3
run st_c2 --rows 1000000 --steps 200 --opt mfma_min_nq=1
python benchmarks/show.py gpurun_out/nq1s/*.json
