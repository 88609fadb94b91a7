# r03: the pipelined tile kernel's filter flow at every k for more than 128 queries (default) against the
# list-mode tile kernel (mfma_filter_k=1000 forces list mode): fp32 10M x 384 and bf16 10M x 768
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/pipe_any_k; mkdir -p $O
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --steps 4 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '| ms/step %.3f QPS %.0f'%(r['ms_per_step'], r['value']), 'kernel avg %.3f ms x %d  %.1f'%(r['roofline']['avg_launch_ms'], r['roofline']['launches'], r['roofline']['achieved']))" || exit 1; }
for nq in 200 256 512 1024; do for k in 1 10 32; do
  run --nq $nq --k $k; run --nq $nq --k $k --opt mfma_filter_k=1000
done; done 2>&1 | tee $O/f32.txt
for nq in 256 1024; do for k in 10 32; do
  run --storage bf16 --dim 768 --nq $nq --k $k; run --storage bf16 --dim 768 --nq $nq --k $k --opt mfma_filter_k=1000
done; done 2>&1 | tee $O/bf16.txt
