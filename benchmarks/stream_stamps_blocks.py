#!/usr/bin/env python3
"""Which blocks of the batch-1 stream kernel finish late, and is it the same ones every launch?  (Diagnostic build,
see benchmarks/stream_stamps.py.)

    RAG_MI355X_LIB=benchmarks/lib_stamps.so python3 benchmarks/stream_stamps_blocks.py <rows>
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
rows = int(sys.argv[1]); dim = 384
import torch
from rag_system_amd import _lib, faiss_compat as faiss
lib = _lib.engine(0)
fn = lib.rag_debug_stream_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
idx = faiss.IndexFlatIP(dim); idx._b.reserve(rows); idx._b.add_synthetic(rows, 1234, 0)
qg = faiss.IndexFlatIP(dim); qg._b.add_synthetic(1, 4321, 0); Q = qg.reconstruct_n(0, 1)
st = torch.cuda.Stream(); q_dev = torch.from_numpy(Q).cuda()
D = torch.zeros(10, dtype=torch.float32, device="cuda"); I = torch.zeros(10, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
buf = np.zeros(1024 * 8, dtype=np.uint64); runs = []
for it in range(25):
    idx._b.search_dev(1, q_dev.data_ptr(), 10, D.data_ptr(), I.data_ptr(), 0, st.cuda_stream); st.synchronize()
    fn(buf.ctypes.data, buf.size); runs.append(buf.reshape(1024, 8)[:256].astype(np.int64).copy())
done = np.array([(r[:, 2] - r[:, 0].min()) / 100.0 for r in runs[5:]])   # launches x blocks
dur = np.array([(r[:, 2] - r[:, 1]) / 100.0 for r in runs[5:]])
m = done.mean(axis=0)
print("rows", rows, "mean finish per block: min %.1f med %.1f max %.1f" % (m.min(), np.median(m), m.max()))
print("launch-to-launch correlation of per-block finish (mean pairwise):", np.mean([np.corrcoef(done[i], done[i+1])[0,1] for i in range(len(done)-1)]))
for x in range(8):
    print("  XCD %d (blocks b%%8==%d): mean finish %.1f  spread %.1f..%.1f" % (x, x, m[x::8].mean(), m[x::8].min(), m[x::8].max()))
o = np.argsort(m); print("slowest blocks:", o[-12:], m[o[-12:]].round(1)); print("fastest blocks:", o[:12], m[o[:12]].round(1))
# within an XCD, by position
print("per-launch spread (max-median):", np.round((done.max(axis=1) - np.median(done, axis=1)), 1))
