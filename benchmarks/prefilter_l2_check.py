#!/usr/bin/env python3
"""IndexFlatL2 through the opt-in prefilter at the headline size (10M x 384 unit rows): queries redone and wall clock
beside the plain search.  python benchmarks/prefilter_l2_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_system_amd import faiss_compat as faiss  # noqa: E402

n, d, k = 10_000_000, 384, 10
idx = faiss.IndexFlatL2(d)
idx._b.add_synthetic(n, 1234, 0)
qg = faiss.IndexFlatIP(d)
qg._b.add_synthetic(1024, 4321, 0)
Qall = qg.reconstruct_n(0, 1024)
for nq in (1, 64, 1024):
    Q = np.ascontiguousarray(Qall[:nq])
    res = {}
    for pf in (False, True):
        idx.set_prefilter(pf)
        idx.search(Q, k)
        t0 = time.perf_counter()
        reps = 20 if nq < 1024 else 3
        for _ in range(reps):
            D, I = idx.search(Q, k)
        res[pf] = ((time.perf_counter() - t0) / reps * 1e3, D, I)
    redone = idx._b.prefilter_redone()
    same = (res[False][2] == res[True][2]).mean()
    print("L2 nq=%d: plain %.3f ms, prefilter %.3f ms, redone %d, ids equal %.4f, max |dD| %.2e"
          % (nq, res[False][0], res[True][0], redone, same, np.abs(res[False][1] - res[True][1]).max()), flush=True)
