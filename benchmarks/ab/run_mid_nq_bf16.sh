#!/bin/bash
# bfloat16 rows, batches just above 64 queries: chunked wide passes against the tile kernel (10M x 768, k = 10)
mkdir -p gpurun_out/midb
for nq in 65 96 128 192; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch-leg --steps 10 --warmup 2 --storage bf16 --dim 768 --nq $nq > gpurun_out/midb/w_nq$nq.json 2> gpurun_out/midb/w_nq$nq.err || echo "nq $nq failed"
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch-leg --steps 10 --warmup 2 --storage bf16 --dim 768 --nq $nq --opt wide_max_nq=-1 > gpurun_out/midb/t_nq$nq.json 2> gpurun_out/midb/t_nq$nq.err || echo "nq $nq failed"
done
python benchmarks/show.py gpurun_out/midb/*.json
