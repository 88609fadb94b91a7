#!/bin/bash
# A/B of the register-fed stream kernel (csrc/flat_stream.hip) against the tile / scan kernels for
# small batches over 10M x 384: batch size, k, wave-tile height (stream_rg), blocks per CU.
# Run on the GPU box: bash benchmarks/run_stream_ab.sh
set -o pipefail
mkdir -p gpurun_out/stream
run() {  # name, args...
    local name=$1; shift
    timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 30 --warmup 5 "$@" \
        > gpurun_out/stream/$name.json 2> gpurun_out/stream/$name.err || { echo "$name failed"; tail -3 gpurun_out/stream/$name.err; return 1; }
}
for nq in 2 3 4 8 16; do
    run old_nq$nq --nq $nq --opt stream_max_nq=-1 || exit 1
    for rg in 2 4; do
        for bpc in 3 4; do
            run st_nq${nq}_rg${rg}_b$bpc --nq $nq --opt mfma_min_nq=2 --opt stream_rg=$rg --opt stream_bpc=$bpc || exit 1
        done
    done
done
for nq in 4 16; do
    run old_nq${nq}_k100 --nq $nq --k 100 --opt stream_max_nq=-1 || exit 1
    run st_nq${nq}_k100 --nq $nq --k 100 || exit 1
    run st_nq${nq}_k100_list --nq $nq --k 100 --opt mfma_filter_k=1000 || exit 1
done
for st in bf16; do
    for nq in 2 4 16; do
        run old_${st}_nq$nq --nq $nq --storage $st --dim 768 --opt stream_max_nq=-1 || exit 1
        run st_${st}_nq$nq --nq $nq --storage $st --dim 768 || exit 1
    done
done
python benchmarks/show.py gpurun_out/stream/*.json
