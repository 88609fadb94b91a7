import json, sys
for f in sys.argv[1:]:
    try:
        j = json.load(open(f)); r = j["roofline"]
        print("%-36s %-58s %9.1f QPS %8.3f ms/step  %s %8.1f frac %.3f kernel %.3f ms" % (
            f.split("/")[-1], j["metric"], j["value"], j["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["avg_launch_ms"]))
    except Exception as e:
        print(f, "ERR", e)
