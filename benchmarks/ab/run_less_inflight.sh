#!/bin/bash
# "Less in flight" on the other small-batch launches: deep-ring stream shapes with fewer blocks per CU,
# wide kernel with fewer waves per block.
set -e
mkdir -p gpurun_out/less
run() { tag=$1; shift
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --steps 100 --warmup 10 "$@" > gpurun_out/less/$tag.json
    python3 -c "
import json; d=json.load(open('gpurun_out/less/$tag.json')); print('$tag [$*] ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
for bpc in 0 2 3; do
  run s_k100_b$bpc --k 100 --opt stream_bpc=$bpc
  run s_nq4k50_b$bpc --nq 4 --k 50 --opt stream_bpc=$bpc
  run s_nq3k64_b$bpc --nq 3 --k 64 --opt stream_bpc=$bpc
  run s_nq8_b$bpc --nq 8 --opt stream_bpc=$bpc
  run s_bf16k100_b$bpc --storage bf16 --dim 768 --k 100 --opt stream_bpc=$bpc
done
for wpb in 0 1 2 3; do
  run w_nq16_w$wpb --nq 16 --opt wide_wpb=$wpb
  run w_nq32_w$wpb --nq 32 --opt wide_wpb=$wpb
done
for wpb in 0 4 6 8; do
  run w_nq64_w$wpb --nq 64 --opt wide_wpb=$wpb
done
