#!/bin/bash
# N=4 (and N=2) functional rehearsal of bench.py's N>1 path on one GPU (gloo, packed lists staged through the host).
set -o pipefail
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
RAG_BENCH_REHEARSAL=1 RAG_BENCH_REHEARSAL_TIMEOUT=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 20 --warmup 3 --rows 2000000 --c4-rows 400000 --dump-result gpurun_out/r02/reh4.npz > gpurun_out/r02/reh4_staged.json 2> gpurun_out/r02/reh4_staged.err
rc=$?
echo "rehearsal N=4 rc=$rc"
tail -1 gpurun_out/r02/reh4_staged.json | cut -c1-3000
[ $rc -ne 0 ] && { grep -v "amdgpu.ids\|socket.cpp" gpurun_out/r02/reh4_staged.err | head -30; exit $rc; }
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --rows 2000000 --no-legs --no-cpu-baseline --no-batch-leg --dump-result gpurun_out/r02/reh1.npz > gpurun_out/r02/reh1.json 2> gpurun_out/r02/reh1.err || exit 1
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/r02/reh4.npz"); b=np.load("gpurun_out/r02/reh1.npz")
print("N=4 ids == N=1 ids:", np.array_equal(a["I"],b["I"]), " scores equal:", np.array_equal(a["D"],b["D"]))
assert np.array_equal(a["I"],b["I"]) and np.array_equal(a["D"],b["D"])
PY
