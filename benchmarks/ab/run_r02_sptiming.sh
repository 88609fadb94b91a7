#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
RAG_MI355X_LIB=$PWD/benchmarks/lib_sptiming.so DOCS=1000000 timeout -k 10 200 python benchmarks/sparse_ab.py > gpurun_out/r02/sptiming.txt 2> gpurun_out/r02/sptiming.err
grep "^select k=30" gpurun_out/r02/sptiming.txt | sed -n '20,23p'
grep "^select k=70" gpurun_out/r02/sptiming.txt | sed -n '20,23p'
grep "^rank phase" gpurun_out/r02/sptiming.txt | sed -n '20,24p'
grep "^count loop" gpurun_out/r02/sptiming.txt | sed -n '40,48p'
timeout -k 10 200 python benchmarks/sparse_ab.py 2>/dev/null | tail -3
