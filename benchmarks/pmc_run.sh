# PMC passes (one rocprofv3 run each) over one bench.py configuration.
#   bash benchmarks/pmc_run.sh "<bench args>" <kernel name substring> <output name>
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
ARGS="bench.py --no-cpu-baseline --no-batch-leg --steps 3 --warmup 1 $1"
OUT=gpurun_out/pmc_$3
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/p1 --output-format csv -- python3 $ARGS > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $OUT/p2 --output-format csv -- python3 $ARGS > $OUT/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS -d $OUT/p3 --output-format csv -- python3 $ARGS > $OUT/p3.log 2>&1
python3 benchmarks/pmc_summary.py $OUT $2 > $OUT/summary.json
cat $OUT/summary.json
