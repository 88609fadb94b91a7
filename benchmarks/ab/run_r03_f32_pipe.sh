cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/f32_1024; mkdir -p $O
for tag in list pipe; do
  opts=""; [ $tag = pipe ] && opts="--opt mfma_filter_k=1 --opt mfma_shape=10"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/$tag --output-format csv -- python3 bench.py --no-cpu-baseline --no-legs --no-batch-leg --nq 1024 --steps 4 --warmup 1 $opts > $O/$tag.log 2>&1 || exit 1
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); echo "== $tag"; cut -c1-150 $f | head -8
done
