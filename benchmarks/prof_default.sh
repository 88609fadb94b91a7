#!/bin/bash
# rocprofv3 kernel stats of the exact default command (python3 bench.py), and the JSON line it printed.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/prof_default; rm -rf $O; mkdir -p $O
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
f=$(ls -t $(find $O -name "*kernel_stats.csv") | head -1); cp $f $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/prof_default/kernel_stats.csv")))
for r in rows[:6]:
    print("%-80s calls %6s avg_us %10.1f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
d = json.loads(open("gpurun_out/prof_default/bench.json").read().strip().splitlines()[-1])
print("bench: ms/step %.4f value %.1f roofline %s" % (d["ms_per_step"], d["value"], json.dumps(d["roofline"])))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
PY
