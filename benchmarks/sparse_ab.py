#!/usr/bin/env python3
"""BM25 / TF-IDF call time at config C5 size, fused kernels against the per-term launches, interleaved
rounds in one process; results of both paths compared bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_system_amd import _lib  # noqa: E402
from benchmarks.hybrid_workload import HybridWorkload  # noqa: E402

lib = _lib.engine(0)
nq = 40
wl = HybridWorkload(int(os.environ.get("DOCS", 1_000_000)), 200_000, 384, nq, 10,
                    lambda m: print("[sparse_ab] " + m, file=sys.stderr, flush=True))
cand = np.arange(0, 60, dtype=np.int64) * 997


def run_bm25(qi):
    return wl.bm.top_k(wl.queries[qi], 30)


def run_tfidf(qi):
    tids = np.asarray([wl.vocab_map[t] for t in wl.queries[qi] if t in wl.vocab_map], dtype=np.int64)
    uq, cnt = np.unique(tids, return_counts=True)
    return wl.ts.ranks_from_row(uq, cnt * wl.tfidf_idf[uq], 70, cand)


res = {}
for fused in (1, 0, 1, 0):
    _lib.check(lib.rag_set_option(b"sparse_fused", fused))
    for name, fn in (("bm25", run_bm25), ("tfidf", run_tfidf)):
        for qi in range(3):
            fn(qi)
        t0 = time.perf_counter()
        outs = [fn(qi) for qi in range(nq)]
        el = (time.perf_counter() - t0) / nq * 1e3
        res.setdefault((name, fused), []).append(el)
        key = (name, "out")
        if key in res:
            for a, b in zip(res[key], outs):
                for x, y in zip(a, b):
                    assert np.array_equal(np.asarray(x), np.asarray(y)), (name, fused)
        else:
            res[key] = outs
_lib.check(lib.rag_set_option(b"sparse_fused", 1))
for name in ("bm25", "tfidf"):
    print("%-6s ms/call: fused %s | per-term %s" % (name, ["%.4f" % t for t in res[(name, 1)]], ["%.4f" % t for t in res[(name, 0)]]))
print("results identical between the two paths")
