# C2 (1M x 384 fp32, one query, k = 10): launch shapes of the stream kernel on a small shard
cd "$GRAFT_REPO_ROOT"
for opts in "" "--opt stream_nb=3" "--opt stream_bpc=2" "--opt stream_nb=3 --opt stream_bpc=2" "--opt stream_nb=3 --opt stream_bpc=4" "--opt stream_fuse=-1" "--opt stream_rg=2"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows 1000000 --nq 1 --steps 400 --warmup 20 $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opts', 'ms/step %.4f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.4f ms'%(r['roofline']['avg_launch_ms']), 'pipelined %.0f'%r['pipelined']['value'])" || exit 1
done
