#!/bin/bash
# r03: rocprofv3 kernel stats + trace of config C5 (one-call tail)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r03/prof_hybrid
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 benchmarks/bench_hybrid.py --queries 60 --no-cpu "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
f=$(ls -t $(find $OUT -name "*kernel_stats.csv") | head -1)
cp "$f" gpurun_out/r03/hybrid_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-80s calls %6s avg_us %10.1f pct %5s" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
t=$(ls -t $(find $OUT -name "*kernel_trace.csv") | head -1)
python3 - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last query: print the last 12 kernels with start offsets
last=rows[-14:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print("%9.1f us +%7.1f  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:70]))
PY
