# rocprofv3 summaries of the opt-in prefilter (bench.py --prefilter): kernel stats for nq 1 / 64 / 1024 and the HBM
# bytes of the nq = 1 pass over the bfloat16 copy (own --pmc pass)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/prof_pf; rm -rf $O; mkdir -p $O
for nq in 1 64 1024; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/nq$nq --output-format csv -- python3 bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq $nq --steps 20 --warmup 3 --prefilter > $O/nq$nq.log 2>&1 || exit 1
  grep -h "^{" $O/nq$nq.log | tail -1 > $O/nq${nq}_under_rocprof.json
done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/nq1_fetch --output-format csv -- python3 bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq 1 --steps 20 --warmup 3 --prefilter > $O/nq1_fetch.log 2>&1 || exit 1
python3 benchmarks/pmc_summary.py $O/nq1_fetch flat_stream_kernel --skip 3 > $O/nq1_fetch.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/nq1_fetch.json | head -30
