cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02/nqprof; rm -rf $O; mkdir -p $O
for nq in 16 32 64; do
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/s$nq --output-format csv -- python3 bench.py --no-cpu-baseline --no-batch-leg --no-legs --nq $nq --steps 20 --warmup 3 > $O/s$nq.log 2>&1 || echo fail $nq
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
