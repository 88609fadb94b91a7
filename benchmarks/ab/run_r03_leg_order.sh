cd "$GRAFT_REPO_ROOT"
p() { python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['c2']['nq1']; print('$1: c2 nq1 ms/step %.4f kernel %.4f'%(c['ms_per_step'], c['roofline']['avg_launch_ms']), '| main kernel %.4f'%r['roofline']['avg_launch_ms'])"; }
timeout -k 10 200 python bench.py --no-cpu-baseline --legs c2 2>/dev/null | p "batch+prefilter legs, then c2"
timeout -k 10 200 python bench.py --no-cpu-baseline --legs c2 --k 32 2>/dev/null | p "batch leg only (k=32: no prefilter leg), then c2"
timeout -k 10 200 python bench.py --no-cpu-baseline --legs c2 --no-batch-leg 2>/dev/null | p "no batch leg, then c2"
