#!/usr/bin/env python3
"""Where a batch-1 step's time beyond its scan kernel goes (10M x 384 fp32, k = 10): the same step
driven five ways, 300 steps each, interleaved rounds in one process."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_system_amd import _lib, faiss_compat as faiss  # noqa: E402
from rag_system_amd.sharded import ShardedFlatSearcher  # noqa: E402

lib = _lib.engine(0)
n, d, k = int(os.environ.get("ROWS", 10_000_000)), 384, 10
dev = torch.device("cuda:0")
idx = faiss.IndexFlatIP(d)
idx._b.add_synthetic(n, 1234, 0)
qg = faiss.IndexFlatIP(d)
qg._b.add_synthetic(1, 4321, 0)
Q = qg.reconstruct_n(0, 1)
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    qd = torch.from_numpy(Q).to(dev)
out = torch.zeros(ShardedFlatSearcher.words(1, k), dtype=torch.int64).pin_memory()
Dp, Ip = out[k:].view(torch.float32)[:k], out[:k]
h = idx._b._h
sh = st.cuda_stream
hip = C.CDLL("libamdhip64.so")
hip.hipStreamSynchronize.argtypes = [C.c_void_p]


def local_search(q, kk, D_loc, I_loc, s):
    idx._b.search_dev(1, q.data_ptr(), kk, D_loc.data_ptr(), I_loc.data_ptr(), 0, s)


searcher = ShardedFlatSearcher(local_search, 0, 1, k, dev, stream=st, out=out)
qp, dp, ip = C.c_void_p(qd.data_ptr()), C.c_void_p(Dp.data_ptr()), C.c_void_p(Ip.data_ptr())
shp = C.c_void_p(sh)
Dh = np.empty((1, k), np.float32)
Ih = np.empty((1, k), np.int64)


def v_searcher():
    searcher.search(qd)
    st.synchronize()


def v_direct():
    lib.rag_flat_search_dev(h, 1, qp, k, dp, ip, 0, shp)
    st.synchronize()


def v_direct_rawsync():
    lib.rag_flat_search_dev(h, 1, qp, k, dp, ip, 0, shp)
    hip.hipStreamSynchronize(shp)


def v_host():
    lib.rag_flat_search(h, 1, _lib.ptr(Q), k, _lib.ptr(Dh), _lib.ptr(Ih))


def v_noalloc():
    lib.rag_flat_search_dev_noalloc(h, 1, qp, k, dp, ip, 0, shp)
    hip.hipStreamSynchronize(shp)


variants = [("searcher+torch sync", v_searcher), ("direct ctypes + torch sync", v_direct),
            ("direct ctypes + raw hipStreamSynchronize", v_direct_rawsync), ("rag_flat_search (host variant)", v_host),
            ("noalloc + raw sync", v_noalloc)]
for name, f in variants:
    for _ in range(20):
        f()
res = {name: [] for name, _ in variants}
for prof in (False, True):
    idx._b.profiling(prof)
    for rnd in range(3):
        for name, f in variants:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                f()
            torch.cuda.synchronize()
            res[name].append(((time.perf_counter() - t0) / 300 * 1e3, prof))
    idx._b.timing()
idx._b.profiling(False)
for name, _ in variants:
    a = [t for t, p in res[name] if not p]
    b = [t for t, p in res[name] if p]
    print("%-45s  ms/step: prof off min %.4f med %.4f | prof on min %.4f med %.4f" % (name, min(a), sorted(a)[1], min(b), sorted(b)[1]))
print(idx._b.last_kernel())
