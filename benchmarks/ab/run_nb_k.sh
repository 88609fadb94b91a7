#!/bin/bash
# Ring depth 2 + one block per CU ("light") against ring 3 + four blocks per CU over k and nq <= 4; tests first.
set -e
mkdir -p gpurun_out/nb_k
timeout -k 10 900 python3 -u -m pytest tests/test_flat_stream_gpu.py tests/test_flat_gpu.py tests/test_flat_bf16_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/nb_k/pytest.log 2>&1 || { tail -30 gpurun_out/nb_k/pytest.log; exit 1; }
tail -2 gpurun_out/nb_k/pytest.log
run() { nb=$1; shift; tag=$(echo "$@" | tr -d ' -')
    timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-batch-leg --steps 200 --warmup 20 --opt stream_nb=$nb "$@" > gpurun_out/nb_k/nb${nb}_$tag.json
    python3 -c "
import json; d=json.load(open('gpurun_out/nb_k/nb${nb}_$tag.json')); print('nb $nb [$*] ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
for args in "--k 10" "--k 20" "--k 32" "--k 48" "--k 64" "--nq 4 --k 10" "--nq 4 --k 20" "--nq 4 --k 32" "--nq 4 --k 100" "--nq 3 --k 64" "--rows 1000000 --k 10" "--rows 1000000 --k 30 --nq 4" "--rows 100000"; do
  run 2 $args
  run 3 $args
done
