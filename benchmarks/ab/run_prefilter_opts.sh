# stream-kernel launch shapes for the prefilter's pass over the bf16 copy (10M x 384, nq = 1 / 8, k = 10)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/prefilter; mkdir -p $O
for nq in 1 8; do
for opts in "" "--opt stream_nb=3" "--opt stream_bpc=2" "--opt stream_bpc=3" "--opt stream_nb=3 --opt stream_bpc=2" "--opt stream_nb=3 --opt stream_bpc=4" "--opt stream_rg=2" "--opt stream_rg=2 --opt stream_bpc=2" "--opt stream_rg=2 --opt stream_nb=3 --opt stream_bpc=2"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq $nq --steps 50 --warmup 3 --prefilter $opts 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nq=$nq $opts', 'ms/step %.3f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.3f ms x %d'%(r['roofline']['avg_launch_ms'], r['roofline']['launches']))" || exit 1
done; done 2>&1 | tee $O/opts.txt
