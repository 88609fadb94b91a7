#!/usr/bin/env python3
"""Where a batch-1 search's microseconds go: per-block time stamps of the stream kernel (diagnostic build).

    bash benchmarks/build_variant.sh stamps flat_stream.hip -DRAG_STREAM_STAMPS
    RAG_MI355X_LIB=benchmarks/lib_stamps.so python3 benchmarks/stream_stamps.py [rows=1000000] [dim=384]

Prints, over the blocks of one launch (median over the launches), the 100 MHz clock at: kernel entry, queries
staged, end of the row loop, list stored, ticket drawn -- each relative to the first block's entry -- and the end
of the fused merge in the last block."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    import torch
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine(0)
    fn = lib.rag_debug_stream_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int]
    idx = faiss.IndexFlatIP(dim)
    idx._b.reserve(rows)
    idx._b.add_synthetic(rows, 1234, 0)
    qg = faiss.IndexFlatIP(dim)
    qg._b.add_synthetic(1, 4321, 0)
    Q = qg.reconstruct_n(0, 1)
    st = torch.cuda.Stream()
    q_dev = torch.from_numpy(Q).cuda()
    D = torch.zeros(10, dtype=torch.float32, device="cuda")
    I = torch.zeros(10, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    runs = []
    for it in range(25):
        idx._b.search_dev(1, q_dev.data_ptr(), 10, D.data_ptr(), I.data_ptr(), 0, st.cuda_stream)
        st.synchronize()
        assert fn(buf.ctypes.data, buf.size) == 0
        runs.append(buf.reshape(1024, 8).copy())
    print("kernel:", idx._b.last_kernel())
    nb = int((runs[-1][:, 0] != 0).sum())
    names = ["entry", "queries staged", "row loop done", "list stored", "ticket drawn"]
    agg = []
    for r in runs[5:]:
        s = r[:nb].astype(np.int64)
        t0 = s[:, 0].min()
        rel = (s[:, :5] - t0) / 100.0          # microseconds
        last = int(np.argmax(s[:, 4]))
        agg.append([rel[:, i].min() for i in range(5)] + [np.median(rel[:, i]) for i in range(5)]
                   + [rel[:, i].max() for i in range(5)] + [(s[last, 5] - t0) / 100.0])
    a = np.median(np.array(agg), axis=0)
    print("%d blocks; microseconds after the first block's entry (median over %d launches)" % (nb, len(agg)))
    for i, n in enumerate(names):
        print("  %-16s first %8.2f   median %8.2f   last %8.2f" % (n, a[i], a[5 + i], a[10 + i]))
    print("  fused merge done (last block)  %8.2f" % a[15])


if __name__ == "__main__":
    main()
