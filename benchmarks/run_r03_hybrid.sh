#!/bin/bash
# r03: config C5 with the one-call tail against the three-call route, and through the product class
set -e
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 500 python3 benchmarks/bench_hybrid.py --queries 60 --search-engine --cpu-sample-docs 40000 --cpu-queries 4 > $O/hyb_tail.json 2> $O/hyb_tail.err
timeout -k 10 300 python3 benchmarks/bench_hybrid.py --queries 60 --no-tail --no-cpu > $O/hyb_notail.json 2> $O/hyb_notail.err
python3 - <<'PY'
import json
for f in ("hyb_tail", "hyb_notail"):
    j = json.load(open("gpurun_out/r03/%s.json" % f))
    print(f, "%.3f ms/query" % j["ms_per_query"], {k: round(v, 4) for k, v in j["stage_ms"].items()})
    if "through_search_engine" in j:
        t = j["through_search_engine"]; print("  product:", "%.3f ms" % t["ms_per_query"], {k: round(v, 4) for k, v in t["stage_ms"].items()})
    if "cpu_baseline" in j: print("  cpu:", j["cpu_baseline"]["value"])
PY
