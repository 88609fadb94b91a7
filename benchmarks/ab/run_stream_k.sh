#!/bin/bash
# Stream kernel, larger k: filter mode (default) against list mode (mfma_filter_k=1000), 10M x 384.
mkdir -p gpurun_out/stream_k
for a in "1 100" "2 100" "4 100" "8 100" "8 64" "4 16" "1 64"; do
    set -- $a
    timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 30 --warmup 5 --nq $1 --k $2 \
        > gpurun_out/stream_k/f_nq$1_k$2.json 2> gpurun_out/stream_k/f_nq$1_k$2.err || exit 1
    timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 30 --warmup 5 --nq $1 --k $2 --opt mfma_filter_k=1000 \
        > gpurun_out/stream_k/l_nq$1_k$2.json 2> gpurun_out/stream_k/l_nq$1_k$2.err || exit 1
done
python benchmarks/show.py gpurun_out/stream_k/*.json
