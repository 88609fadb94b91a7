# the dense stage of config C5 (1M x 384 fp32, one query, k = 90): list mode against the filter flow
cd "$GRAFT_REPO_ROOT"
for rows in 1000000 10000000; do for opts in "" "--opt mfma_filter_k=65" "--opt mfma_filter_k=65 --opt mfma_filter_step=16" "--opt mfma_filter_k=65 --opt mfma_filter_min_rows=1"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows $rows --k 90 --nq 1 --steps 200 --warmup 10 $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows=$rows $opts', 'ms/step %.4f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.4f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
done; done
