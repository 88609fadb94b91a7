cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pmc_merge; rm -rf $O; mkdir -p $O
A="bench.py --no-cpu-baseline --k 100 --steps 10 --warmup 2"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/p1 --output-format csv -- python3 $A > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/p2 --output-format csv -- python3 $A > $O/p2.log 2>&1
python3 benchmarks/pmc_summary.py $O flat_merge_kernel --skip 2
