#!/bin/bash
# sample stride of the filter flow for large batches (config C3 and fp32 nq=1024 k=100)
mkdir -p gpurun_out/fstep
for st in 32 64 128; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --storage bf16 --dim 768 --k 100 --nq 1024 --steps 10 --warmup 2 --opt mfma_filter_step=$st > gpurun_out/fstep/c3_s$st.json 2> gpurun_out/fstep/c3_s$st.err || echo "c3 $st failed"
    timeout -k 10 200 python bench.py --no-cpu-baseline --k 100 --nq 1024 --steps 5 --warmup 1 --opt mfma_filter_step=$st > gpurun_out/fstep/f32_s$st.json 2> gpurun_out/fstep/f32_s$st.err || echo "f32 $st failed"
done
python benchmarks/show.py gpurun_out/fstep/*.json
