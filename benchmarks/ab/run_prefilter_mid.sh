# the prefilter's pass over the bf16 copy for mid batches (10M x 384, k = 10): ring depth of the wide kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/prefilter; mkdir -p $O
for nq in 16 32 48 64; do
for opts in "--opt wide_nb=4" "--opt wide_nb=3"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq $nq --steps 50 --warmup 3 --prefilter $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nq=$nq $opts', 'ms/step %.3f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.3f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
done; done 2>&1 | tee $O/mid.txt
