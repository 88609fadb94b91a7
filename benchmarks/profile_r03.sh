# Round-3 profiles: kernel-trace stats and PMC passes, each in its own rocprofv3 run.
#   gpurun --timeout 1100 -- bash benchmarks/profile_r03.sh ; then python3 benchmarks/collect_profiles_r03.py
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/prof_r03; rm -rf $O; mkdir -p $O
run() { name=$1; shift
    timeout -s KILL 400 rocprofv3 "$@" > $O/$name.log 2>&1 || echo "$name failed"; }
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
NQ1="bench.py --no-cpu-baseline --no-batch-leg --no-legs --steps 50 --warmup 5"
C3="bench.py --no-cpu-baseline --no-legs --storage bf16 --dim 768 --k 100 --nq 1024 --steps 3 --warmup 1"
# 1. exactly the default command
run default_stats --kernel-trace --stats -d $O/default_stats --output-format csv -- python3 bench.py
# 1b. the headline alone (every launch of the batch-1 kernel has the same size: its average is comparable
#     with roofline.avg_launch_ms of the same run)
run nq1_stats --kernel-trace --stats -d $O/nq1_stats --output-format csv -- python3 $NQ1
# 2. HBM traffic of the batch-1 kernel
run nq1_fetch --pmc FETCH_SIZE -d $O/nq1_fetch --output-format csv -- python3 $NQ1
run nq1_write --pmc WRITE_SIZE -d $O/nq1_write --output-format csv -- python3 $NQ1
# 3. C3 with the pipelined tile kernel
run c3_stats --kernel-trace --stats -d $O/c3_stats --output-format csv -- python3 $C3
run c3_sq --pmc $SQ -d $O/c3_sq --output-format csv -- python3 $C3
run c3_grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $O/c3_grbm --output-format csv -- python3 $C3
run c3_fetch --pmc FETCH_SIZE -d $O/c3_fetch --output-format csv -- python3 $C3
# 3b. the batch-1024 half of the headline (fp32 rows through the pipelined tile kernel's filter flow)
B1024="bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq 1024 --steps 4 --warmup 1"
run b1024_stats --kernel-trace --stats -d $O/b1024_stats --output-format csv -- python3 $B1024
run b1024_sq --pmc $SQ -d $O/b1024_sq --output-format csv -- python3 $B1024
run b1024_grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $O/b1024_grbm --output-format csv -- python3 $B1024
# 4. hybrid (C5)
run hyb_stats --kernel-trace --stats -d $O/hyb_stats --output-format csv -- python3 benchmarks/bench_hybrid.py --no-cpu --queries 60
for d in nq1_fetch nq1_write; do python3 benchmarks/pmc_summary.py $O/$d flat_stream_kernel --skip 5 > $O/$d.json; done
for d in c3_sq c3_grbm c3_fetch b1024_sq b1024_grbm; do python3 benchmarks/pmc_summary.py $O/$d flat_pipe_kernel --skip 1 > $O/$d.json; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
grep -h "^{" $O/default_stats.log | tail -1 > $O/default_under_rocprof.json
grep -h "^{" $O/nq1_stats.log | tail -1 > $O/nq1_under_rocprof.json
grep -h "^{" $O/c3_stats.log | tail -1 > $O/c3_under_rocprof.json
grep -h "^{" $O/hyb_stats.log | tail -1 > $O/hyb_under_rocprof.json
grep -h "^{" $O/b1024_stats.log | tail -1 > $O/b1024_under_rocprof.json
ls -R $O | head -60
# 5. the default line once more WITHOUT the profiler (the numbers DESIGN.md quotes)
timeout -k 10 600 python3 bench.py > $O/default_plain.json 2> $O/default_plain.err || echo "plain default failed"
