# r03: launch shapes of the stream kernel on one rank's shard of config 4 (12.5M x 768 fp32, nq 1, k 10); the default
# first, in the middle and last (the first run on a fresh box is ~5 % faster than any later one, whatever the options)
cd "$GRAFT_REPO_ROOT"
for o in "" "--opt stream_nb=3" "" "--opt stream_bpc=2" "--opt stream_rg=2" ""; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows 12500000 --dim 768 --steps 40 --warmup 5 $o 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$o] ms/step %.4f kernel avg %.4f ms %.1f GB/s'%(r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['achieved']))" || exit 1
done
