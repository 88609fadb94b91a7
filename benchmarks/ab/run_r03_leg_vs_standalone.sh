cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch-leg --legs c2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['c2']['nq1']; print('leg after 10M main: ms/step %.4f kernel %.4f'%(c['ms_per_step'], c['roofline']['avg_launch_ms'])); c=r['c2']['nq64']; print('  nq64 leg: ms/step %.4f kernel %.4f'%(c['ms_per_step'], c['roofline']['avg_launch_ms']))"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch-leg --rows 2000000 --legs c2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['c2']['nq1']; print('leg after 2M main: ms/step %.4f kernel %.4f'%(c['ms_per_step'], c['roofline']['avg_launch_ms']))"
timeout -k 10 100 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows 1000000 --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('standalone: ms/step %.4f kernel %.4f'%(r['ms_per_step'], r['roofline']['avg_launch_ms']))"
timeout -k 10 100 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows 1000000 --nq 64 --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('standalone nq64: ms/step %.4f kernel %.4f'%(r['ms_per_step'], r['roofline']['avg_launch_ms']))"
