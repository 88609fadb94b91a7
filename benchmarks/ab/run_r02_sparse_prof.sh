#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_sparse" -o sp -- python3 "$GRAFT_REPO_ROOT/benchmarks/sparse_ab.py" > "$GRAFT_REPO_ROOT/gpurun_out/r02/sparse_ab_prof.txt" 2> "$GRAFT_REPO_ROOT/gpurun_out/r02/sparse_ab_prof.err"
echo rc=$?
cd "$GRAFT_REPO_ROOT"
find gpurun_out/r02/prof_sparse -name "*kernel_stats*" | head
f=$(find gpurun_out/r02/prof_sparse -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-220 "$f" | head -30
