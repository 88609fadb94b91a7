#!/usr/bin/env python3
"""Copy the summaries produced by benchmarks/profile_r01.sh (gpurun_out/prof_r01) into profiles/
and derive the per-kernel figures DESIGN.md quotes.  Usage: python benchmarks/collect_profiles.py r01"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = "gpurun_out/prof_r01/"


def ld(n):
    return json.load(open(P + n + ".json"))


def dominant(j, counter=None):
    """The kernel instantiation with the largest value of `counter` (default: any first counter)."""
    best, bv = None, -1.0
    for name, cs in j.items():
        c = counter if counter in cs else sorted(cs)[0]
        v = cs[c]["mean"] * cs[c]["n"]
        if v > bv:
            best, bv = name, v
    return best, j[best]


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


for src, dst in (("nq1_stats", "bench_nq1"), ("nq1024_stats", "bench_nq1024"), ("c3_stats", "bench_c3_bf16_nq1024_k100"),
                 ("hyb_stats", "bench_hybrid_1m"), ("nq4_stats", "bench_nq4"), ("nq64_stats", "bench_nq64")):
    shutil.copy(newest(P + src + "/*/*_kernel_stats.csv"), "profiles/%s_%s_kernel_stats.csv" % (tag, dst))

name, f = dominant(ld("nq1_fetch"))
_, w = dominant(ld("nq1_write"))
json.dump({"kernel": name, "FETCH_SIZE_KiB_units": f["FETCH_SIZE"], "WRITE_SIZE_KiB_units": w["WRITE_SIZE"],
           "note": "separate --pmc passes; FETCH_SIZE on gfx950 reports half of the streamed bytes "
                   "(MI355X_MICROARCH.md HBM/rocprofv3 section): HBM read bytes = value*1024*2"},
          open("profiles/%s_bench_nq1_pmc_hbm.json" % tag, "w"), indent=1)
hbm = f["FETCH_SIZE"]["mean"] * 1024 * 2
json.dump({"rows": 10000000, "dim": 384, "nq": 1, "storage": "f32", "hbm_bytes_per_launch": hbm,
           "source": "profiles/%s_bench_nq1_pmc_hbm.json: rocprofv3 --pmc FETCH_SIZE (own pass), KiB units x2 gfx950 "
                     "correction, mean over the timed launches of the batch-1 kernel (flat_stream_kernel)" % tag}, open("profiles/traffic.json", "w"), indent=1)

out = {}
for key, names in (("f32_nq1024_k10", ("nq1024_sq", "nq1024_grbm", "nq1024_fetch")),
                   ("bf16_768_nq1024_k100_filter_mode", ("c3_sq", "c3_grbm")),
                   ("f32_nq4_k10_stream_kernel", ("nq4_sq", "nq4_grbm", "nq4_fetch")),
                   ("f32_nq64_k10_wide_kernel_filter_pass", ("nq64_sq", "nq64_grbm", "nq64_fetch"))):
    d = {}
    for n in names:
        kname, cs = dominant(ld(n), "SQ_WAVE_CYCLES" if n.endswith("_sq") else None)
        d.setdefault("kernels", {})[n] = kname
        d.update({k: v["mean"] for k, v in cs.items()})
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    d["derived"] = {"cycles_per_xcd": cyc, "mfma_util": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                    "wait_any_frac": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                    "wait_inst_any_frac": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]}
    if "FETCH_SIZE" in d:
        d["derived"]["hbm_read_bytes"] = d["FETCH_SIZE"] * 1024 * 2
    out[key] = d
json.dump(out, open("profiles/%s_mfma_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out.items()}, indent=1), hbm)
