# needs the diagnostic build first: bash benchmarks/build_variant.sh stamps flat_stream.hip -DRAG_STREAM_STAMPS
# r03: the tail of a batch-1 search: GPU tests of the stream kernel, C2 / headline / small-corpus timings, block stamps
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -u -m pytest tests/test_flat_stream_gpu.py tests/test_flat_gpu.py tests/test_zz_no_alloc.py tests/test_configs_gpu.py tests/test_robustness_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
line() { python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ms/step %.4f QPS %.1f'%(r['ms_per_step'], r['value']), 'kernel avg %.4f ms x %d  %.1f GB/s'%(r['roofline']['avg_launch_ms'], r['roofline']['launches'], r['roofline']['achieved']), 'pipelined', r.get('pipelined',{}).get('value'))"; }
for i in 1 2; do
timeout -k 10 100 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows 1000000 --steps 300 --warmup 20 2>/dev/null | line "c2" || exit 1
timeout -k 10 100 python bench.py --no-cpu-baseline --no-legs --no-batch-leg 2>/dev/null | line "headline" || exit 1
done
timeout -k 10 100 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows 20000 --steps 500 --warmup 20 2>/dev/null | line "20k rows" || exit 1
RAG_MI355X_LIB=benchmarks/lib_stamps.so timeout -k 5 120 python3 benchmarks/stream_stamps.py 10000000 384 || exit 1
RAG_MI355X_LIB=benchmarks/lib_stamps.so timeout -k 5 120 python3 benchmarks/stream_stamps.py 1000000 384
