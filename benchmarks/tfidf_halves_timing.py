#!/usr/bin/env python3
"""Wall clock of the two halves of the TF-IDF query (rag_tfidf_query_begin / _end) against the one call, on the
C5 workload (1M documents).  python benchmarks/tfidf_halves_timing.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.hybrid_workload import HybridWorkload  # noqa: E402

wl = HybridWorkload(1_000_000, 200_000, 384, 40, 10, lambda m: None)
rng = np.random.default_rng(1)
tb = te = to = 0.0
n = 0
for rep in range(3):
    for qi in range(40):
        raw = wl.queries[qi]
        tids = np.asarray([wl.vocab_map[t] for t in raw if t in wl.vocab_map], dtype=np.int64)
        uq, cnt = np.unique(tids, return_counts=True)
        w = cnt * wl.tfidf_idf[uq]
        cand = rng.integers(0, wl.n, 40).astype(np.int64)
        t0 = time.perf_counter()
        tok = wl.ts.ranks_from_row_begin(uq, w, 70)
        t1 = time.perf_counter()
        time.sleep(0.0003)                      # the dense pass and the BM25 scoring would run here
        t2 = time.perf_counter()
        a = wl.ts.ranks_from_row_end(tok, 50, cand)
        t3 = time.perf_counter()
        b = wl.ts.ranks_from_row(uq, w, 50, cand)
        t4 = time.perf_counter()
        assert a[0].tolist() == b[0].tolist() and a[2].tolist() == b[2].tolist()
        if rep:
            tb += t1 - t0
            te += t3 - t2
            to += t4 - t3
            n += 1
print("begin %.1f us, end %.1f us, one call %.1f us" % (tb / n * 1e6, te / n * 1e6, to / n * 1e6))
