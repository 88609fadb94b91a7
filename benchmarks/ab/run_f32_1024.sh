# the fp32 half of the headline (10M x 384, nq 1024, k 10): list mode (default) against the filter flow, tile shapes 4 and 10
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/f32_1024; mkdir -p $O
for opts in "" "--opt mfma_filter_k=1" "--opt mfma_filter_k=1 --opt mfma_shape=10" "--opt mfma_filter_k=1 --opt mfma_shape=9" "--opt mfma_filter_k=1 --opt mfma_shape=8"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq 1024 --steps 5 --warmup 2 $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opts', 'ms/step %.3f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.3f ms x %d  %.1f TF'%(r['roofline']['avg_launch_ms'], r['roofline']['launches'], r['roofline']['achieved']))" || exit 1
done 2>&1 | tee $O/ab.txt
