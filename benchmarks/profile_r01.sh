# Round-1 profiles: kernel-trace stats and PMC passes, each in its own rocprofv3 run.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
# PART=1: batch-1 / batch-1024 / C3 runs, PART=2: the register-fed kernels and the hybrid bench
# (two gpurun calls of <= 20 minutes each; gpurun_out/prof_r01 of part 1 must still be present
# locally for collect_profiles.py, part 2 only adds files)
PART=${PART:-1}
O=gpurun_out/prof_r01; [ "$PART" = 1 ] && [ -z "$ONLY" ] && rm -rf $O; mkdir -p $O
run() { name=$1; shift
    case $name in nq4_*|nq64_*|hyb_*) p=2;; *) p=1;; esac
    [ "$p" = "$PART" ] || return 0
    if [ -n "$ONLY" ]; then case $name in ${ONLY}*) ;; *) return 0;; esac; fi   # ONLY=nq1_: rerun one group (PART=1 keeps the rest)
    timeout -s KILL 240 rocprofv3 "$@" > $O/$name.log 2>&1 || echo "$name failed"; }
NQ1="bench.py --no-cpu-baseline --no-batch-leg --steps 50 --warmup 5"
NQ1024="bench.py --no-cpu-baseline --nq 1024 --steps 3 --warmup 1"
C3="bench.py --no-cpu-baseline --storage bf16 --dim 768 --k 100 --nq 1024 --steps 3 --warmup 1"
run nq1_stats --kernel-trace --stats -d $O/nq1_stats --output-format csv -- python3 $NQ1
run nq1_fetch --pmc FETCH_SIZE -d $O/nq1_fetch --output-format csv -- python3 $NQ1
run nq1_write --pmc WRITE_SIZE -d $O/nq1_write --output-format csv -- python3 $NQ1
run nq1024_stats --kernel-trace --stats -d $O/nq1024_stats --output-format csv -- python3 $NQ1024
run nq1024_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/nq1024_sq --output-format csv -- python3 $NQ1024
run nq1024_grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $O/nq1024_grbm --output-format csv -- python3 $NQ1024
run nq1024_fetch --pmc FETCH_SIZE -d $O/nq1024_fetch --output-format csv -- python3 $NQ1024
run c3_stats --kernel-trace --stats -d $O/c3_stats --output-format csv -- python3 $C3
run c3_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/c3_sq --output-format csv -- python3 $C3
run c3_grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $O/c3_grbm --output-format csv -- python3 $C3
NQ4="bench.py --no-cpu-baseline --no-batch-leg --nq 4 --steps 20 --warmup 3"
NQ64="bench.py --no-cpu-baseline --no-batch-leg --nq 64 --steps 10 --warmup 2"
run nq4_stats --kernel-trace --stats -d $O/nq4_stats --output-format csv -- python3 $NQ4
run nq4_fetch --pmc FETCH_SIZE -d $O/nq4_fetch --output-format csv -- python3 $NQ4
run nq4_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/nq4_sq --output-format csv -- python3 $NQ4
run nq4_grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $O/nq4_grbm --output-format csv -- python3 $NQ4
run nq64_stats --kernel-trace --stats -d $O/nq64_stats --output-format csv -- python3 $NQ64
run nq64_fetch --pmc FETCH_SIZE -d $O/nq64_fetch --output-format csv -- python3 $NQ64
run nq64_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/nq64_sq --output-format csv -- python3 $NQ64
run nq64_grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $O/nq64_grbm --output-format csv -- python3 $NQ64
run hyb_stats --kernel-trace --stats -d $O/hyb_stats --output-format csv -- python3 benchmarks/bench_hybrid.py --no-cpu --queries 20
if [ "$PART" = 1 ]; then
for d in nq1_fetch nq1_write; do python3 benchmarks/pmc_summary.py $O/$d flat_stream_kernel --skip 5 > $O/$d.json; done
[ -n "$ONLY" ] || for d in nq1024_sq nq1024_grbm nq1024_fetch c3_sq c3_grbm; do python3 benchmarks/pmc_summary.py $O/$d flat_mfma_kernel --skip 1 > $O/$d.json; done
else
for d in nq4_fetch nq4_sq nq4_grbm; do python3 benchmarks/pmc_summary.py $O/$d flat_stream_kernel --skip 1 > $O/$d.json; done
for d in nq64_fetch nq64_sq nq64_grbm; do python3 benchmarks/pmc_summary.py $O/$d "flat_wide_kernel<4, 0, false, false>" --skip 1 > $O/$d.json; done
fi
# keep only the summaries (traces are large)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -R $O | head -60
