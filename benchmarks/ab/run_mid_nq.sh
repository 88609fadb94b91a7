#!/bin/bash
# tile kernel for batches just above the wide kernel's range (10M x 384 fp32, k = 10)
mkdir -p gpurun_out/mid
for nq in 65 96 128 192 256 512; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch-leg --steps 10 --warmup 2 --nq $nq > gpurun_out/mid/nq$nq.json 2> gpurun_out/mid/nq$nq.err || echo "nq $nq failed"
done
python benchmarks/show.py gpurun_out/mid/*.json
