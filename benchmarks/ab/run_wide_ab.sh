#!/bin/bash
# 16x16 register-fed kernel (csrc/flat_wide.hip) against the tile kernel, 10M x 384 fp32 and 10M x 768 bf16.
mkdir -p gpurun_out/wide
run() { local name=$1; shift
    timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 30 --warmup 5 "$@" \
        > gpurun_out/wide/$name.json 2> gpurun_out/wide/$name.err || { echo "$name failed"; tail -3 gpurun_out/wide/$name.err; exit 1; }; }
for nq in 17 32 48 64; do
    run old_nq$nq --nq $nq --opt wide_max_nq=-1
    run w_nq$nq --nq $nq
    run w_nq${nq}_k100 --nq $nq --k 100
    run old_nq${nq}_k100 --nq $nq --k 100 --opt wide_max_nq=-1
done
for nq in 12 16; do
    run w_nq$nq --nq $nq --opt wide_min_nq=1
    run w_nq${nq}_k100 --nq $nq --k 100 --opt wide_min_nq=1
done
for nq in 32 64; do
    run old_bf16_nq$nq --nq $nq --storage bf16 --dim 768 --opt wide_max_nq=-1
    run w_bf16_nq$nq --nq $nq --storage bf16 --dim 768
done
python benchmarks/show.py gpurun_out/wide/*.json
