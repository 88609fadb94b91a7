"""GPU parity tests of the register-fed small-batch kernel (csrc/flat_stream.hip): batches of
1..16 queries, k <= 128, list mode and filter mode.

Checked against the float64 oracle (oracle/dense.py, restating the IndexFlat.search call at
vector_storage.py:196) to the fp32 bar of BASELINE.json (1e-5 relative, ids identical
except near-ties), and against the tile kernel it replaces for these batch sizes.
"""
import numpy as np
import pytest

from oracle import dense as od

pytestmark = pytest.mark.gpu

RTOL = 1e-5
ATOL = 2e-6


def _rand_unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    return x


def _scorer(X, Q, metric):
    X64, Q64 = X.astype(np.float64), Q.astype(np.float64)

    def f(qi, ids):
        rows = X64[ids]
        return rows @ Q64[qi] if metric == 0 else ((rows - Q64[qi]) ** 2).sum(axis=1)
    return f


def _opts(lib, **kw):
    from rag_system_amd import _lib
    for name, val in kw.items():
        _lib.check(lib.rag_set_option(name.encode(), val))


def _check(idx, X, Q, k, metric, storage):
    rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
    Xr, Qr = rnd(X), rnd(Q)
    D, I = idx.search(Q, k)
    _, Ir, D64 = od.flat_search(Xr, Qr, k, metric)
    od.check_topk(D, I, D64, Ir, RTOL, ATOL if metric == 0 else 4e-6, _scorer(Xr, Qr, metric))
    return D, I


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d", [(1, 384), (7, 64), (16, 384), (17, 40), (31, 768), (300, 384), (4099, 768), (3001, 37),
                                 (20_011, 384), (1000, 20), (513, 8), (2000, 1000), (50_003, 72), (200, 4096)])
def test_list_mode_matches_oracle(engine, n, d, metric, storage):
    """d = 37, 20, 8, 72, 1000: rows that end inside a 256-byte segment; n = 16, 17, 31, 513: a single
    or a window-shifted last wave tile (n = 1, 7: below the kernel's 16 rows, served by the tile
    kernel); d = 4096: the longest fp32 rows whose queries still fit the kernel's LDS budget."""
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(n * 11 + d + metric)
    X = _rand_unit(rng, n, d)
    if n > 100:
        X[n // 2] = X[3]
        X[n - 1] = X[3]
    idx = faiss.IndexFlat(d, metric, storage=storage)
    idx.add(X)
    for nq, k in [(3, 10), (4, 1), (5, 64), (8, 33), (9, 100), (12, 128), (16, 10), (16, 65), (2, 10), (1, 10), (1, 100), (2, 128)]:
        Q = _rand_unit(rng, nq, d)
        Q[0] = X[min(3, n - 1)]
        _check(idx, X, Q, k, metric, storage)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_stream_kernel_agrees_with_tile_kernel_and_tile_heights(engine, storage):
    """Same ids as the MFMA tile kernel (stream_max_nq = -1) and for both wave-tile heights; the
    scores differ only in summation order."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(17)
    X = _rand_unit(rng, 30_011, 200)
    idx = faiss.IndexFlat(200, 0, storage=storage)
    idx.add(X)
    try:
        for nq, k in [(3, 10), (6, 50), (13, 100), (16, 20)]:
            Q = _rand_unit(rng, nq, 200)
            D0, I0 = _check(idx, X, Q, k, 0, storage)
            for rg in (2, 4):
                for bpc in (1, 3, 5):
                    _opts(lib, stream_rg=rg, stream_bpc=bpc)
                    D1, I1 = _check(idx, X, Q, k, 0, storage)
                    assert np.array_equal(I0, I1), (nq, k, rg, bpc)
                    assert np.allclose(D0, D1, rtol=1e-6, atol=1e-7)
            _opts(lib, stream_rg=0, stream_bpc=0, stream_max_nq=-1)
            D2, I2 = _check(idx, X, Q, k, 0, storage)
            _opts(lib, stream_max_nq=0)
            assert np.array_equal(I0, I2), (nq, k)
            assert np.allclose(D0, D2, rtol=2e-6, atol=2e-7)
    finally:
        _opts(lib, stream_rg=0, stream_bpc=0, stream_max_nq=0)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("metric", [0, 1])
def test_short_ring_shapes_match_oracle_and_deep_ring(engine, metric, storage):
    """The light shapes (up to 4 queries, k <= 64) are built with a ring of 2 segment buffers and
    run one block per CU (option stream_nb: 0 = by rule, 2 / 3 forced): same ids as the ring of 3,
    for rows that end inside a segment, an odd number of segments, and several blocks per CU."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(4242 + metric)
    try:
        for n, d in [(16, 384), (300, 64), (5003, 37), (20_011, 384), (9_000, 200), (3001, 1000), (70_001, 72)]:
            X = _rand_unit(rng, n, d)
            if n > 100:
                X[n // 2] = X[3]
                X[n - 1] = X[3]
            idx = faiss.IndexFlat(d, metric, storage=storage)
            idx.add(X)
            for nq, k in [(1, 10), (1, 64), (2, 33), (3, 1), (4, 16), (4, 64)]:
                Q = _rand_unit(rng, nq, d)
                Q[0] = X[min(3, n - 1)]
                _opts(lib, stream_nb=3, stream_bpc=0)
                D3, I3 = _check(idx, X, Q, k, metric, storage)
                for bpc in (0, 1, 3):
                    _opts(lib, stream_nb=2, stream_bpc=bpc)
                    D2, I2 = _check(idx, X, Q, k, metric, storage)
                    assert np.array_equal(I2, I3), (n, d, nq, k, bpc)
                    assert np.allclose(D2, D3, rtol=2e-6, atol=2e-7)
        # filter mode with the short ring in the filter pass and in the sample pass
        _opts(lib, stream_nb=2, stream_bpc=0, mfma_filter_k=1, mfma_filter_min_rows=1)
        X = _rand_unit(rng, 70_001, 72)
        idx = faiss.IndexFlat(72, metric, storage=storage)
        idx.add(X)
        for nq, k in [(1, 10), (4, 100), (3, 50)]:
            Q = _rand_unit(rng, nq, 72)
            _check(idx, X, Q, k, metric, storage)
    finally:
        _opts(lib, stream_nb=0, stream_bpc=0, mfma_filter_k=0, mfma_filter_min_rows=1 << 19)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_filter_mode_matches_oracle(engine, storage):
    """Sample pass and filter pass by the stream kernel (redo by the tile kernel), forced on
    small inputs, including row orders that defeat the 1-in-32 sample."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(777)
    rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
    try:
        _opts(lib, mfma_filter_k=1, mfma_filter_min_rows=1)
        for metric in (0, 1):
            for n, d in [(300, 64), (20_011, 384), (70_001, 72), (9_000, 37)]:
                X = _rand_unit(rng, n, d)
                X[n // 2] = X[3]
                X[n - 1] = X[3]
                idx = faiss.IndexFlat(d, metric, storage=storage)
                idx.add(X)
                for nq, k in [(3, 10), (5, 1), (8, 100), (16, 50), (11, 128)]:
                    Q = _rand_unit(rng, nq, d)
                    Q[0] = X[3]
                    _check(idx, X, Q, k, metric, storage)
        n, d, k = 40_000, 32, 20
        Q = _rand_unit(rng, 12, d)
        base = _rand_unit(rng, n, d) * 0.01
        tile = np.arange(n) // 256
        cases = {}
        a = base.copy(); a[:] = a[0]
        cases["all_equal"] = a
        b = base.copy(); b[(tile % 32 == 0)] *= 100.0
        cases["best_in_sample"] = b
        c = base.copy(); c[(tile % 32 == 5)] *= 100.0
        cases["best_outside_sample"] = c
        e = base.copy(); e[:300] = Q[0] * 3.0
        cases["ties_at_top"] = e
        for rows in (8, 16):                      # the stream kernel samples every 32nd wave tile of 8 / 16 rows
            wt = np.arange(n) // rows
            f = base.copy(); f[(wt % 32 == 0)] *= 100.0
            cases["best_in_sample_%d" % rows] = f
            g = base.copy(); g[(wt % 32 == 5)] *= 100.0
            cases["best_outside_sample_%d" % rows] = g
        for name, X in cases.items():
            idx = faiss.IndexFlat(d, 0, storage=storage)
            idx.add(X)
            Xr, Qr = rnd(X), rnd(Q)
            D, I = idx.search(Q, k)
            _, Ir, D64 = od.flat_search(Xr, Qr, k, 0)
            od.check_topk(D, I, D64, Ir, RTOL, ATOL, _scorer(Xr, Qr, 0))
            if name in ("all_equal", "ties_at_top"):
                assert np.array_equal(I[0], Ir[0]), name
    finally:
        _opts(lib, mfma_filter_k=0, mfma_filter_min_rows=1 << 19)


def test_default_thresholds_on_a_large_shard(engine):
    """2M x 96 rows: k = 10 runs list mode, k = 100 filter mode by default; both exact, and the
    ids do not depend on which kernel made the pass."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    n, d, seed = 2_000_000, 96, 99
    idx = faiss.IndexFlat(d, 0)
    idx._b.add_synthetic(n, seed, 0)
    Q = od.synth_queries(5, 16, d)
    Q[1] = od.synth_rows(seed, 1_234_567, 1, d)[0]
    for nq, k in [(16, 10), (16, 100), (4, 100), (7, 64)]:
        D, I = idx.search(Q[:nq], k)
        _, Ir, D64 = od.flat_search_synth(seed, 0, n, d, Q[:nq], k)

        def scorer(qi, ids):
            rows = np.stack([od.synth_rows(seed, int(i), 1, d)[0] for i in ids])
            return rows.astype(np.float64) @ Q[qi].astype(np.float64)
        od.check_topk(D, I, D64, Ir, RTOL, ATOL, scorer)
        assert I[1, 0] == 1_234_567
        try:
            _opts(lib, stream_max_nq=-1)
            D2, I2 = idx.search(Q[:nq], k)
        finally:
            _opts(lib, stream_max_nq=0)
        assert np.array_equal(I, I2), (nq, k)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("metric", [0, 1])
def test_fused_merge_equals_merge_kernel(engine, metric, storage):
    """1-2 queries, k <= 16, in list mode: the stream kernel's last block merges the per-block lists
    itself (stream_fused_merge); larger k keeps the merge kernel.  Bit-identical to the separate flat_merge_kernel launch (option
    stream_fuse = -1) and exact against the oracle: few and many blocks, k from 1 to 128, k larger
    than the number of rows and of lists (padding), a non-zero id base, and tie runs long enough to
    take the k-way merge path (all-equal rows; 700 copies of the best row)."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(4242 + metric)
    rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
    cases = []
    for n, d in [(16, 64), (150, 384), (5_000, 96), (70_001, 64), (300_000, 32)]:
        X = _rand_unit(rng, n, d)
        cases.append(("random", X))
    Xe = np.tile(_rand_unit(rng, 1, 48), (40_000, 1))          # every score equal: ties decide everything
    cases.append(("all_equal", Xe))
    Xt = _rand_unit(rng, 60_000, 48)
    Xt[rng.choice(60_000, 700, replace=False)] = Xt[5]          # 700 exact copies of one row
    cases.append(("ties_at_top", Xt))
    try:
        for name, X in cases:
            n, d = X.shape
            idx = faiss.IndexFlat(d, metric, storage=storage)
            idx.add(X)
            Xr = rnd(X)
            for nq, k in [(1, 1), (1, 10), (2, 10), (2, 16), (1, 3), (1, 64), (2, 100), (1, 128), (2, 17)]:
                Q = _rand_unit(rng, nq, d)
                Q[0] = X[5]
                Qr = rnd(Q)
                _opts(lib, stream_fuse=0)
                D, I = idx.search(Q, k)
                fused = "fused-merge" in idx._b.last_kernel()
                _opts(lib, stream_fuse=-1)
                D0, I0 = idx.search(Q, k)
                assert "fused-merge" not in idx._b.last_kernel()
                assert np.array_equal(I, I0) and np.array_equal(D, D0), (name, n, d, nq, k)
                assert fused == (n >= 16 and k <= 16), (name, n, nq, k, idx._b.last_kernel())
                _, Ir, D64 = od.flat_search(Xr, Qr, k, metric)
                od.check_topk(D, I, D64, Ir, RTOL, ATOL if metric == 0 else 4e-6, _scorer(Xr, Qr, metric))
                if name != "random":
                    assert np.array_equal(I, Ir), (name, nq, k)   # exact ties: lowest ids first, like the oracle
        # id base through the device entry point
        import torch
        X = _rand_unit(rng, 20_000, 64)
        idx = faiss.IndexFlat(64, metric, storage=storage)
        idx.add(X)
        Q = _rand_unit(rng, 2, 64)
        _opts(lib, stream_fuse=0)
        D, I = idx.search(Q, 10)
        qd = torch.from_numpy(Q).cuda()
        Dd = torch.empty((2, 10), dtype=torch.float32, device="cuda")
        Id = torch.empty((2, 10), dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        idx._b.search_dev(2, qd.data_ptr(), 10, Dd.data_ptr(), Id.data_ptr(), 7_000_000_000, st)
        torch.cuda.synchronize()
        assert "fused-merge" in idx._b.last_kernel()
        assert np.array_equal(Id.cpu().numpy(), I + 7_000_000_000) and np.array_equal(Dd.cpu().numpy(), D)
    finally:
        _opts(lib, stream_fuse=0)


def test_fused_merge_never_reads_a_stale_list(engine):
    """The last block of the stream kernel merges the block lists the other blocks wrote in the same launch; the
    hand-off (write-through stores, a ticket, one acquire) leans on hardware behaviour outside the HIP memory
    model.  This test makes a stale read VISIBLE instead of lucky: one handle whose
    workspace is reused by corpora of different size (300k, 20k and 2k rows: 256, 256 and fewer block lists)
    and by queries with disjoint answers, many searches enqueued back to back on one stream with no host
    synchronisation between them, every result compared with the separate merge kernel's.  A list left over
    from the previous launch -- another corpus, another query -- would surface as a foreign id or score."""
    import ctypes as C
    import torch
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(99)
    d, k, nrep = 64, 10, 12
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(dev)
    corpora = []
    for n in (300_000, 20_000, 2_000):
        X = _rand_unit(rng, n, d)
        Q = _rand_unit(rng, 6, d)
        for j in range(6):                   # each query has its own planted neighbourhood, far from the others'
            X[rng.choice(n, 12, replace=False)] = (Q[j] + 0.05 * _rand_unit(rng, 12, d)).astype(np.float32)
        corpora.append((X, Q))
    idx = faiss.IndexFlatIP(d)
    try:
        for rnd in range(6):
            X, Q = corpora[rnd % 3]
            idx.reset()
            idx.add(X)
            _opts(lib, stream_fuse=-1)
            ref = [idx.search(Q[j:j + 1], k) for j in range(6)]          # merge kernel (also flushes the rows)
            assert "fused-merge" not in idx._b.last_kernel()
            _opts(lib, stream_fuse=0)
            with torch.cuda.stream(st):
                qd = torch.from_numpy(Q).to(dev)
                Dd = torch.full((nrep * 6, k), -7.0, dtype=torch.float32, device=dev)
                Id = torch.full((nrep * 6, k), -7, dtype=torch.int64, device=dev)
            st.synchronize()
            for r in range(nrep):             # 72 launches back to back, alternating queries, no host sync
                for j in range(6):
                    slot = r * 6 + j
                    _lib.check(lib.rag_flat_search_dev(idx._b._h, 1, C.c_void_p(qd[j].data_ptr()), k,
                                                       C.c_void_p(Dd[slot].data_ptr()), C.c_void_p(Id[slot].data_ptr()), 0,
                                                       C.c_void_p(st.cuda_stream)), lib)
            assert "fused-merge" in idx._b.last_kernel(), idx._b.last_kernel()
            st.synchronize()
            Dh, Ih = Dd.cpu().numpy(), Id.cpu().numpy()
            for r in range(nrep):
                for j in range(6):
                    D0, I0 = ref[j]
                    assert np.array_equal(Ih[r * 6 + j], I0[0]), (rnd, r, j)
                    assert np.array_equal(Dh[r * 6 + j].view(np.uint32), D0[0].view(np.uint32)), (rnd, r, j)
    finally:
        _opts(lib, stream_fuse=0)
