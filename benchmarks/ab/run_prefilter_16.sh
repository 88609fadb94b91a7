cd "$GRAFT_REPO_ROOT"
for nq in 12 16; do for opts in "" "--opt wide_min_nq=17"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq $nq --steps 50 --warmup 3 --prefilter $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nq=$nq $opts', 'ms/step %.3f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.3f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
done; done
