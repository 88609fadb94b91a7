#!/bin/bash
# fp32 rows of 768 values (config C4's row length), 5M rows: batch sizes around the wide kernel's LDS limit
mkdir -p gpurun_out/d768
for nq in 16 32 48 64 96 128 192; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch-leg --rows 5000000 --dim 768 --steps 10 --warmup 2 --nq $nq $EXTRA > gpurun_out/d768/${TAG:-cur}_nq$nq.json 2> gpurun_out/d768/${TAG:-cur}_nq$nq.err || echo "nq $nq failed"
done
python benchmarks/show.py gpurun_out/d768/${TAG:-cur}_*.json
