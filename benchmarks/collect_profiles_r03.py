#!/usr/bin/env python3
"""Copy the summaries produced by benchmarks/profile_r03.sh (gpurun_out/prof_r03) into profiles/
(r02_*) and derive the figures DESIGN.md quotes.  Usage: python benchmarks/collect_profiles_r03.py"""
import glob
import json
import os
import shutil

P, tag = "gpurun_out/prof_r03/", "r03"


def ld(n):
    return json.load(open(P + n + ".json"))


def only(j):
    name = max(j, key=lambda k: sum(c["mean"] * c["n"] for c in j[k].values()))
    return name, j[name]


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


for src, dst in (("default_stats", "bench_default"), ("nq1_stats", "bench_nq1"), ("c3_stats", "bench_c3_bf16_nq1024_k100"), ("hyb_stats", "bench_hybrid_1m"), ("b1024_stats", "bench_nq1024")):
    shutil.copy(newest(P + src + "/*/*_kernel_stats.csv"), "profiles/%s_%s_kernel_stats.csv" % (tag, dst))
for src, dst in (("default_under_rocprof", "bench_default_under_rocprof"), ("nq1_under_rocprof", "bench_nq1_under_rocprof"), ("c3_under_rocprof", "bench_c3_under_rocprof"),
                 ("hyb_under_rocprof", "bench_hybrid_1m_under_rocprof"), ("b1024_under_rocprof", "bench_nq1024_under_rocprof")):
    shutil.copy(P + src + ".json", "profiles/%s_%s.json" % (tag, dst))

if os.path.exists(P + "default_plain.json") and os.path.getsize(P + "default_plain.json"):
    shutil.copy(P + "default_plain.json", "profiles/%s_bench_default.json" % tag)

name, f = only(ld("nq1_fetch"))
_, w = only(ld("nq1_write"))
json.dump({"kernel": name, "FETCH_SIZE_KiB_units": f["FETCH_SIZE"], "WRITE_SIZE_KiB_units": w["WRITE_SIZE"],
           "note": "separate --pmc passes; FETCH_SIZE on gfx950 reports half of the streamed bytes "
                   "(MI355X_MICROARCH.md HBM/rocprofv3 section): HBM read bytes = value*1024*2"},
          open("profiles/%s_bench_nq1_pmc_hbm.json" % tag, "w"), indent=1)
hbm = f["FETCH_SIZE"]["mean"] * 1024 * 2
json.dump({"rows": 10000000, "dim": 384, "nq": 1, "storage": "f32", "hbm_bytes_per_launch": hbm,
           "source": "profiles/%s_bench_nq1_pmc_hbm.json: rocprofv3 --pmc FETCH_SIZE (own pass), KiB units x2 gfx950 "
                     "correction, mean over the timed launches of the batch-1 kernel (flat_stream_kernel with the fused "
                     "merge)" % tag}, open("profiles/traffic.json", "w"), indent=1)

d = {"kernels": {}}
for n in ("c3_sq", "c3_grbm", "c3_fetch"):
    kname, cs = only(ld(n))
    d["kernels"][n] = kname
    d.update({k: v["mean"] for k, v in cs.items()})
cyc = d["GRBM_GUI_ACTIVE"] / 8
stats = {r.split(",")[0].strip('"')[:40]: r for r in open("profiles/%s_bench_c3_bf16_nq1024_k100_kernel_stats.csv" % tag)}
d["derived"] = {"cycles_per_xcd": cyc, "mfma_busy_frac": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                "wait_any_frac": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                "wait_inst_any_frac": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                "hbm_read_bytes": d["FETCH_SIZE"] * 1024 * 2,
                "note": "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16) summed over the 1024 SIMDs; "
                        "clock = cycles_per_xcd / the kernel's duration in the kernel stats of the same workload"}
out = {"bf16_768_nq1024_k100_pipelined_kernel_filter_pass": d}
print(json.dumps(d["derived"], indent=1))

# the fp32 half of the headline (10M x 384, nq 1024, k 10) through the same kernel (32x32x2 fp32 tiles)
if os.path.exists(P + "b1024_sq.json"):
    f = {"kernels": {}}
    for n in ("b1024_sq", "b1024_grbm"):
        kname, cs = only(ld(n))
        f["kernels"][n] = kname
        f.update({k: v["mean"] for k, v in cs.items()})
    cyc = f["GRBM_GUI_ACTIVE"] / 8
    f["derived"] = {"cycles_per_xcd": cyc, "mfma_busy_frac": f["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                    "wait_any_frac": f["SQ_WAIT_ANY"] / f["SQ_WAVE_CYCLES"],
                    "wait_inst_any_frac": f["SQ_WAIT_INST_ANY"] / f["SQ_WAVE_CYCLES"],
                    "note": "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per v_mfma_f32_32x32x2_f32) summed over the 1024 SIMDs"}
    out["f32_384_nq1024_k10_pipelined_kernel_filter_pass"] = f
    print(json.dumps(f["derived"], indent=1))
json.dump(out, open("profiles/%s_mfma_pmc.json" % tag, "w"), indent=1)

# agreement of the two clocks on the batch-1 kernel (same run)
import csv
r = json.load(open("profiles/%s_bench_nq1_under_rocprof.json" % tag))
rows = [x for x in csv.DictReader(open("profiles/%s_bench_nq1_kernel_stats.csv" % tag)) if "flat_stream_kernel" in x["Name"]]
print("batch-1 kernel: rocprofv3 average %.4f ms over %s launches; HIP events in the same run %.4f ms over %d launches"
      % (float(rows[0]["AverageNs"]) / 1e6, rows[0]["Calls"], r["roofline"]["avg_launch_ms"], r["roofline"]["launches"]))
