#!/bin/bash
# wide kernel ablation: normal library against lib_wide_nomfma.so (loads + LDS reads only)
mkdir -p gpurun_out/wab
for nq in ${NQS:-16 32 64}; do
  for lib in ${LIBS:-default wide_nomfma}; do
    if [ $lib = default ]; then unset RAG_MI355X_LIB; else export RAG_MI355X_LIB=$PWD/benchmarks/lib_$lib.so; fi
    timeout -k 10 150 python bench.py --no-cpu-baseline --no-batch-leg --steps 30 --warmup 5 --nq $nq --opt wide_min_nq=1 \
        > gpurun_out/wab/${lib}_nq$nq.json 2> gpurun_out/wab/${lib}_nq$nq.err || { echo "$lib $nq failed"; tail -3 gpurun_out/wab/${lib}_nq$nq.err; }
  done
done
python benchmarks/show.py gpurun_out/wab/*.json
