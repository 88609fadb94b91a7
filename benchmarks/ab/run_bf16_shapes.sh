#!/bin/bash
# bfloat16 tile kernel: block shape by batch size above the wide kernel's range (10M x 768, k = 10)
mkdir -p gpurun_out/bshape
for nq in 96 128 192 256 512; do
  for sh in 0 2 3 4 8 9; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch-leg --steps 8 --warmup 2 --storage bf16 --dim 768 --nq $nq --opt wide_max_nq=-1 --opt mfma_shape=$sh > gpurun_out/bshape/nq${nq}_s$sh.json 2> gpurun_out/bshape/nq${nq}_s$sh.err || echo "nq $nq shape $sh failed"
  done
done
python benchmarks/show.py gpurun_out/bshape/*.json
