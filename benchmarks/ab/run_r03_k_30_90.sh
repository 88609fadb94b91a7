cd "$GRAFT_REPO_ROOT"
for rows in 20000 1000000 10000000; do for k in 10 30 90; do
timeout -k 10 100 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows $rows --k $k --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows %d k %d ms/step %.4f kernel avg %.4f ms'%(r['config']['rows'], r['config']['k'], r['ms_per_step'], r['roofline']['avg_launch_ms']))" || exit 1
done; done
