#!/bin/bash
# Round 2, first GPU call: full GPU test suite, default bench, one N=4 rehearsal of the N>1 path on gloo (device path).
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread --durations=15 > gpurun_out/r02/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/r02/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err || { tail -20 gpurun_out/r02/bench_default.err; exit 1; }
cat gpurun_out/r02/bench_default.json | cut -c1-1500
timeout -k 10 200 python bench.py --out device --no-legs --no-cpu-baseline > gpurun_out/r02/bench_outdevice.json 2> gpurun_out/r02/bench_outdevice.err || { tail -20 gpurun_out/r02/bench_outdevice.err; exit 1; }
cut -c1-400 gpurun_out/r02/bench_outdevice.json
RAG_BENCH_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 20 --warmup 3 --rows 2000000 --c4-rows 400000 > gpurun_out/r02/reh4_device.json 2> gpurun_out/r02/reh4_device.err
echo "rehearsal rc=$?"
tail -3 gpurun_out/r02/reh4_device.json | cut -c1-1200
