# bf16 rows, 65-128 queries: the tile kernel of the batch size against the pipelined kernel (256-query tile, mostly empty)
cd "$GRAFT_REPO_ROOT"
for nq in 80 100 128; do for opts in "" "--opt mfma_shape=10"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq $nq --steps 30 --warmup 3 --prefilter $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('384 copy nq=$nq $opts', 'ms/step %.3f'%(r['ms_per_step']), 'kernel avg %.3f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --storage bf16 --dim 768 --k 100 --nq $nq --steps 20 --warmup 3 $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('768 bf16 k100 nq=$nq $opts', 'ms/step %.3f'%(r['ms_per_step']), 'kernel avg %.3f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
done; done
