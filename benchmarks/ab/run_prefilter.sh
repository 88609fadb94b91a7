# A/B of the opt-in bfloat16 prefilter on the headline corpus (10M x 384 fp32, k=10)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/prefilter; mkdir -p $O
for nq in 1 8 16 64 256 1024; do
  st=$((nq >= 256 ? 5 : 50))
  for pf in "" "--prefilter"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq $nq --steps $st --warmup 3 $pf 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nq=$nq $pf', 'ms/step %.3f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.3f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
  done
done 2>&1 | tee $O/ab.txt
