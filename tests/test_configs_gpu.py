"""BASELINE.json configs at their full sizes, through the C ABI, against the streaming oracle
(oracle/flat_ref.c regenerates the synthetic rows on the host; the device generator is checked
bit-identical in tests/test_flat_gpu.py).

  C2   1M x 384 fp32, k = 10, nq = 1 and nq = 64
  C4   one rank's shard of 100M x 768 fp32 at 8 GPUs: 12.5M x 768 generated with
       first_global_row = 7 * 12.5M and searched with id_base = the same (global ids), plus the
       merge of two such shards' lists
  headline, second half: 10M x 384 fp32, nq = 1024, k = 10 (three planted queries)
C3 (10M x 768 bf16) is in tests/test_flat_bf16_gpu.py, C5 in tests/test_hybrid_full_gpu.py.

Bar: scores within 1e-5 relative (+2e-6 absolute) of the float64 oracle, ids identical except
near-ties within that tolerance (re-scored in float64).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import dense as od

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 2e-6
SEED, QSEED = 1234, 4321


def _synth_scorer(first, d, Q, extra=None):
    """float64 scores of arbitrary GLOBAL ids (rows regenerated one by one)."""
    def f(qi, ids):
        rows = np.stack([od.synth_rows(SEED, int(i), 1, d)[0] for i in ids])
        return rows.astype(np.float64) @ Q[qi].astype(np.float64)
    return f


def test_c2_1m_x_384_nq1_and_nq64(engine):
    from rag_system_amd import faiss_compat as faiss
    n, d, k = 1_000_000, 384, 10
    idx = faiss.IndexFlatIP(d)
    idx._b.add_synthetic(n, SEED, 0)
    Q = od.synth_queries(QSEED, 64, d)
    Q[5] = od.synth_rows(SEED, 777_777, 1, d)[0]          # a query equal to a corpus row: score 1.0 on top
    _, Ir, D64 = od.flat_search_synth(SEED, 0, n, d, Q, k)
    assert Ir[5, 0] == 777_777
    D1, I1 = idx.search(Q[:1], k)                         # nq = 1: stream kernel
    od.check_topk(D1, I1, D64[:1], Ir[:1], RTOL, ATOL, _synth_scorer(0, d, Q))
    D, I = idx.search(Q, k)                               # nq = 64: 16x16 register-fed kernel
    od.check_topk(D, I, D64, Ir, RTOL, ATOL, _synth_scorer(0, d, Q))
    assert I[5, 0] == 777_777
    # every single query alone agrees with its row of the batch
    for qi in (5, 63):
        Dq, Iq = idx.search(Q[qi:qi + 1], k)
        od.check_topk(Dq, Iq, D64[qi:qi + 1], Ir[qi:qi + 1], RTOL, ATOL,
                      lambda _q, ids, qi=qi: _synth_scorer(0, d, Q)(qi, ids))


def test_c4_one_rank_shard_12_5m_x_768_with_id_base(engine):
    """Rank 7 of 8 of the 100M x 768 corpus (38.4 GB of rows): generated with its global row
    numbers, searched on the device with id_base, compared with the oracle's GLOBAL ids; then its
    list merged with rank 0's first rows' list exactly as the all-gather + merge step does."""
    import torch
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    per, d, k = 12_500_000, 768, 10
    first = 7 * per
    idx = faiss.IndexFlatIP(d)
    idx._b.reserve(per)
    idx._b.add_synthetic(per, SEED, first)
    assert idx.ntotal == per
    nq = 2
    Q = od.synth_queries(QSEED, nq, d)
    Q[1] = od.synth_rows(SEED, first + per - 1, 1, d)[0]   # the shard's (and the corpus') very last row
    _, Ir, D64 = od.flat_search_synth(SEED, first, per, d, Q, k)
    assert Ir.min() >= 0 and Ir[1, 0] == per - 1          # the oracle numbers rows from 0: local ids
    Ig = Ir + first
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        qd = torch.from_numpy(Q).to(dev)
        Dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
        Id = torch.empty((nq, k), dtype=torch.int64, device=dev)
        idx._b.search_dev(nq, qd.data_ptr(), k, Dd.data_ptr(), Id.data_ptr(), first, side.cuda_stream)
        # one query alone (the batch-1 shard scan the C4 bench leg times)
        D1 = torch.empty((1, k), dtype=torch.float32, device=dev)
        I1 = torch.empty((1, k), dtype=torch.int64, device=dev)
        idx._b.search_dev(1, qd.data_ptr(), k, D1.data_ptr(), I1.data_ptr(), first, side.cuda_stream)
        side.synchronize()

    def scorer(qi, ids):
        assert (np.asarray(ids) >= first).all() and (np.asarray(ids) < first + per).all()
        return _synth_scorer(first, d, Q)(qi, ids)
    od.check_topk(Dd.cpu().numpy(), Id.cpu().numpy(), D64, Ig, RTOL, ATOL, scorer)
    od.check_topk(D1.cpu().numpy(), I1.cpu().numpy(), D64[:1], Ig[:1], RTOL, ATOL, scorer)
    assert int(Id[1, 0]) == first + per - 1

    # a second shard (rank 0's first 300k rows, id_base 0) and the merge of the two lists
    n0 = 300_000
    idx0 = faiss.IndexFlatIP(d)
    idx0._b.add_synthetic(n0, SEED, 0)
    _, Ir0, D640 = od.flat_search_synth(SEED, 0, n0, d, Q, k)
    with torch.cuda.stream(side):
        Dl = torch.empty((2, nq, k), dtype=torch.float32, device=dev)
        Il = torch.empty((2, nq, k), dtype=torch.int64, device=dev)
        idx0._b.search_dev(nq, qd.data_ptr(), k, Dl[0].data_ptr(), Il[0].data_ptr(), 0, side.cuda_stream)
        Dl[1].copy_(Dd)
        Il[1].copy_(Id)
        Dm = torch.empty((nq, k), dtype=torch.float32, device=dev)
        Im = torch.empty((nq, k), dtype=torch.int64, device=dev)
        _lib.check(lib.rag_topk_merge_dev(0, 2, nq, k, C.c_void_p(Dl.data_ptr()), C.c_void_p(Il.data_ptr()),
                                          C.c_void_p(Dm.data_ptr()), C.c_void_p(Im.data_ptr()),
                                          C.c_void_p(side.cuda_stream)), lib)
        side.synchronize()
    # expected merge, from the oracle's two lists (float64 scores rounded like the engine's output)
    De, Ie = od.topk_merge(0, np.stack([D640.astype(np.float32), D64.astype(np.float32)]), np.stack([Ir0, Ig]))

    def scorer2(qi, ids):
        return _synth_scorer(0, d, Q)(qi, ids)
    od.check_topk(Dm.cpu().numpy(), Im.cpu().numpy(), De.astype(np.float64), Ie, RTOL, ATOL, scorer2)
    # and the device merge is bit-identical to the CPU merge of the device's own lists
    Dr, Irr = od.topk_merge(0, Dl.cpu().numpy(), Il.cpu().numpy())
    assert np.array_equal(Irr, Im.cpu().numpy()) and np.array_equal(Dr, Dm.cpu().numpy())


def test_headline_second_half_10m_x_384_nq1024(engine):
    """The batch-1024 half of the headline metric at full size: three checked queries (first, middle,
    last of the batch; one of them a corpus row) inside 1021 filler queries; every row of the result
    is additionally range- and order-checked."""
    from rag_system_amd import faiss_compat as faiss
    n, d, k, nq = 10_000_000, 384, 10, 1024
    idx = faiss.IndexFlatIP(d)
    idx._b.add_synthetic(n, SEED, 0)
    Q3 = od.synth_queries(QSEED, 3, d)
    Q3[2] = od.synth_rows(SEED, 9_999_999, 1, d)[0]
    B = od.synth_queries(999, nq, d)
    pos = [0, nq // 2, nq - 1]
    for j, p in enumerate(pos):
        B[p] = Q3[j]
    D, I = idx.search(B, k)
    assert D.shape == (nq, k) and (I >= 0).all() and (I < n).all() and np.all(D[:, :-1] >= D[:, 1:])
    _, Ir, D64 = od.flat_search_synth(SEED, 0, n, d, Q3, k)
    sel = np.asarray(pos)
    od.check_topk(D[sel], I[sel], D64, Ir, RTOL, ATOL, _synth_scorer(0, d, Q3))
    assert I[nq - 1, 0] == 9_999_999
    # the same queries alone (stream kernel) give the same ids
    D1, I1 = idx.search(Q3[:1], k)
    assert np.array_equal(I1[0], I[0])
