# rocprofv3 kernel stats of one bench.py configuration: bash benchmarks/prof_one.sh <name> <bench args...>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
name=$1; shift
OUT=gpurun_out/prof_$name
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 bench.py --no-cpu-baseline --no-batch-leg --steps 20 --warmup 3 "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
f=$(ls -t $(find $OUT -name "*kernel_stats.csv") | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-90s calls %6s avg_us %10.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
