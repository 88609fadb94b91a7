#!/usr/bin/env python3
"""Would batch-1 searches that overlap on two streams run at the median block's rate?  Two indexes over the same
synthetic rows (a handle has ONE workspace, so two searches of one handle are ordered one behind the other), one stream
each, searches enqueued alternately with `depth` in flight per stream; aggregate QPS against one stream alone.

    python3 benchmarks/two_stream_overlap.py [rows=10000000] [steps=200]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    import torch
    from rag_system_amd import _lib, faiss_compat as faiss
    _lib.engine(0)
    d, k, depth = 384, 10, 2
    idx = []
    for _ in range(2):
        ix = faiss.IndexFlatIP(d)
        ix._b.reserve(rows)
        ix._b.add_synthetic(rows, 1234, 0)
        idx.append(ix)
    qg = faiss.IndexFlatIP(d)
    qg._b.add_synthetic(1, 4321, 0)
    q = torch.from_numpy(qg.reconstruct_n(0, 1)).cuda()
    for prio in (None, (0, -1)):
        if prio is None:
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        else:
            streams = [torch.cuda.Stream(priority=prio[0]), torch.cuda.Stream(priority=prio[1])]
        outs = [[(torch.zeros(k, dtype=torch.float32, device="cuda"), torch.zeros(k, dtype=torch.int64, device="cuda"))
                 for _ in range(depth)] for _ in range(2)]
        evs = [[torch.cuda.Event() for _ in range(depth)] for _ in range(2)]

        def run(n, which):
            for i in range(n):
                for s in which:
                    slot = i % depth
                    if i >= depth:
                        evs[s][slot].synchronize()
                    D, I = outs[s][slot]
                    idx[s]._b.search_dev(1, q.data_ptr(), k, D.data_ptr(), I.data_ptr(), 0, streams[s].cuda_stream)
                    evs[s][slot].record(streams[s])
            torch.cuda.synchronize()

        for which, name in (((0,), "one stream"), ((0, 1), "two streams")):
            run(20, which)
            t0 = time.perf_counter()
            run(steps, which)
            el = time.perf_counter() - t0
            nsearch = steps * len(which)
            print("%s%s: %d searches in %.1f ms = %.4f ms per search = %.1f QPS aggregate"
                  % (name, "" if prio is None else " (priorities %s)" % (prio,), nsearch, el * 1e3, el / nsearch * 1e3, nsearch / el))
        assert torch.equal(outs[0][0][1], outs[1][0][1])


if __name__ == "__main__":
    main()
