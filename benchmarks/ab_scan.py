#!/usr/bin/env python3
"""Interleaved A/B of engine options for the batch-1 scan in ONE process (cdna guide rule 24):
   python benchmarks/ab_scan.py "scan_pipe=0" "scan_pipe=1,scan_blocks=512" ...
Each variant is a comma list of rag_set_option name=value; unspecified options are reset to
their defaults before every variant.  Prints median / min kernel time and GB/s per variant."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from rag_system_amd import _lib, faiss_compat as faiss  # noqa: E402

DEFAULTS = {"scan_blocks": 0, "scan_nt": 1, "scan_rows": 0, "mfma_min_nq": 0, "mfma_shape": 0, "mfma_bpc": 0, "mfma_stages": 0}


def main():
    variants = [v for v in sys.argv[1:] if "=" in v] or ["scan_blocks=0"]
    rows = int(os.environ.get("ROWS", "10000000"))
    d = int(os.environ.get("DIM", "384"))
    nq = int(os.environ.get("NQ", "1"))
    k = int(os.environ.get("K", "10"))
    rounds = int(os.environ.get("ROUNDS", "7"))
    reps = int(os.environ.get("REPS", "10"))
    lib = _lib.engine()
    idx = faiss.IndexFlatIP(d)
    idx._b.add_synthetic(rows, 1234, 0)
    qg = faiss.IndexFlatIP(d)
    qg._b.add_synthetic(nq, 4321, 0)
    Q = qg.reconstruct_n(0, nq)
    res = {v: [] for v in variants}
    ref = None
    for r in range(rounds + 1):
        for v in variants:
            for name, val in DEFAULTS.items():
                _lib.check(lib.rag_set_option(name.encode(), val))
            for kv in v.split(","):
                name, val = kv.split("=")
                _lib.check(lib.rag_set_option(name.encode(), int(val)))
            idx._b.profiling(True)
            idx._b.timing()
            for _ in range(reps):
                D, I = idx.search(Q, k)
            n, ms, nbytes = idx._b.timing()
            idx._b.profiling(False)
            if ref is None:
                ref = I.copy()
            assert np.array_equal(ref, I), "variant %s changed the result ids" % v
            if r > 0:
                res[v].append(ms / n)
    for v in variants:
        med, mn = statistics.median(res[v]), min(res[v])
        print("%-40s median %.4f ms (%.0f GB/s)  min %.4f ms (%.0f GB/s)"
              % (v, med, nbytes / med / 1e6, mn, nbytes / mn / 1e6), flush=True)


if __name__ == "__main__":
    main()
