#!/bin/bash
# r03: `python3 bench.py --gpus 2` launched bare (the driver's SCALE command shape) as a one-GPU rehearsal,
# ids compared with the N=1 dump; then the default line.
set -e
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-batch-leg --no-cpu-baseline --dump-result $O/n1.npz > $O/n1.json 2> $O/n1.err
RAG_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-legs --no-batch-leg --dump-result $O/n2.npz > $O/n2.json 2> $O/n2.err
python3 - <<'PY'
import json, numpy as np
a, b = np.load("gpurun_out/r03/n1.npz"), np.load("gpurun_out/r03/n2.npz")
assert (a["I"] == b["I"]).all() and (a["D"] == b["D"]).all()
j = json.load(open("gpurun_out/r03/n2.json"))
assert j["n_gpus"] == 2 and len(j["stages"]["per_rank"]) == 2
print("N=2 self-launched rehearsal identical to N=1;", j["n_gpus"], "ranks")
PY
