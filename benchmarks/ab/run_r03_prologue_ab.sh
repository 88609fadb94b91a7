cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do for v in oldpro new; do
lib=rag_system_amd/librag_mi355x.so; [ $v = oldpro ] && lib=benchmarks/lib_oldpro.so
for rows in 20000 1000000; do
RAG_MI355X_LIB=$lib timeout -k 10 100 python bench.py --no-cpu-baseline --no-legs --no-batch-leg --rows $rows --steps 400 --warmup 20 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rows %d ms/step %.4f kernel avg %.4f ms'%(r['config']['rows'], r['ms_per_step'], r['roofline']['avg_launch_ms']))" || exit 1
done; done; done
