#!/bin/bash
# r03: the fused merge hand-off: default (relaxed ticket + sc1 stores) against the model-conformant acq_rel ticket
# first: bash benchmarks/build_variant.sh fuse_strict flat_stream.hip -DRAG_FUSE_STRICT
set -e
bash benchmarks/ab/run_lib_ab.sh benchmarks/lib_fuse_strict.so --no-legs
bash benchmarks/ab/run_lib_ab.sh benchmarks/lib_fuse_strict.so --no-legs --rows 1000000
timeout -k 10 300 python -u -m pytest tests/test_flat_stream_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
