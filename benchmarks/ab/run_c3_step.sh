# C3 (10M x 768 bf16, nq 1024, k 100): filter-pass time against the number of appended candidates (sample stride)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c3step; mkdir -p $O
for stp in 16 32 64 128 256; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --storage bf16 --dim 768 --k 100 --nq 1024 --steps 4 --warmup 1 --opt mfma_filter_step=$stp 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step=$stp', 'ms/step %.3f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.3f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
done 2>&1 | tee $O/step.txt
