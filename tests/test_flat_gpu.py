"""GPU parity tests of the dense hot path: HIP (through the C ABI) vs the CPU oracle.

Bar (BASELINE.json north_star): scores within 1e-5 relative of the float64 oracle, ids
identical except ties within that tolerance; integer/index work (padding, tie order,
synthetic rows, merge) bit-exact.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import dense as od

pytestmark = pytest.mark.gpu

RTOL = 1e-5   # stated by BASELINE.json for fp32
ATOL = 2e-6   # absolute floor for scores near zero (|q|.|x| <= ~1 here; fp32 dot error ~1e-7)


def _index(d, metric=0):
    from rag_system_amd import faiss_compat as faiss
    return faiss.IndexFlat(d, metric)


def _rand_unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    return x


def _scorer(X, Q, metric):
    X64 = X.astype(np.float64)
    Q64 = Q.astype(np.float64)

    def f(qi, ids):
        rows = X64[ids]
        if metric == 0:
            return rows @ Q64[qi]
        return ((rows - Q64[qi]) ** 2).sum(axis=1)
    return f


def _check(index, X, Q, k, metric):
    D, I = index.search(Q, k)
    Dr, Ir, D64 = od.flat_search(X, Q, k, metric)
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (Q.shape[0], k)
    return od.check_topk(D, I, D64, Ir, RTOL, ATOL, _scorer(X, Q, metric))


def test_synthetic_rows_bit_identical(engine):
    for d, n, seed, first in [(384, 1000, 1234, 0), (37, 513, 7, 123456789), (768, 64, 99, 10_000_000_000)]:
        idx = _index(d)
        idx._b.add_synthetic(n, seed, first)
        got = idx.reconstruct_n(0, n)
        ref = od.synth_rows(seed, first, n, d)
        assert got.tobytes() == ref.tobytes(), "device generator differs from oracle/flat_ref.c:synth_row"
        nrm = np.linalg.norm(got.astype(np.float64), axis=1)
        assert np.allclose(nrm, 1.0, atol=1e-6)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d", [(1, 384), (7, 384), (1000, 384), (4099, 768), (3001, 37), (257, 4),
                                 (50_003, 384), (1025, 1024), (300, 2050)])
@pytest.mark.parametrize("kernel", ["default", "scan"])
def test_search_matches_oracle(engine, n, d, metric, kernel):
    """kernel = "scan": "mfma_min_nq" = 1000 keeps every batch on the GEMV scan kernel (by default
    it serves k > 128, long rows and shards under 16 rows; smaller k goes to the stream kernel)."""
    from rag_system_amd import _lib
    lib = _lib.engine()
    rng = np.random.default_rng(n * 31 + d)
    X = _rand_unit(rng, n, d)
    idx = _index(d, metric)
    idx.add(X)
    assert idx.ntotal == n
    try:
        if kernel == "scan":
            _lib.check(lib.rag_set_option(b"mfma_min_nq", 1000))
        for nq, k in [(1, 10), (2, 30), (3, 1), (5, 90), (19, 10), (1, 100), (4, 200), (1, 256)]:
            Q = _rand_unit(rng, nq, d)
            _check(idx, X, Q, k, metric)
    finally:
        lib.rag_set_option(b"mfma_min_nq", 0)


def test_unnormalised_rows_and_large_magnitudes(engine):
    # the saved index of the reference holds raw model outputs: no unit-norm assumption
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((5000, 384)) * rng.uniform(0.1, 30, (5000, 1))).astype(np.float32)
    Q = rng.standard_normal((3, 384)).astype(np.float32) * 4
    for metric in (0, 1):
        idx = _index(384, metric)
        idx.add(X)
        D, I = idx.search(Q, 10)
        _, Ir, D64 = od.flat_search(X, Q, 10, metric)
        od.check_topk(D, I, D64, Ir, RTOL, 1e-4, _scorer(X, Q, metric))


def test_k_larger_than_ntotal_pads(engine):
    rng = np.random.default_rng(1)
    X = _rand_unit(rng, 5, 384)
    Q = _rand_unit(rng, 2, 384)
    for metric, pad in ((0, -np.finfo(np.float32).max), (1, np.finfo(np.float32).max)):
        idx = _index(384, metric)
        idx.add(X)
        D, I = idx.search(Q, 10)
        assert (I[:, 5:] == -1).all() and (D[:, 5:] == pad).all()
        assert sorted(I[0, :5].tolist()) == [0, 1, 2, 3, 4]
        _check(idx, X, Q, 10, metric)


def test_empty_index_returns_pads(engine):
    idx = _index(384)
    D, I = idx.search(np.zeros((2, 384), np.float32), 3)
    assert (I == -1).all() and (D == -np.finfo(np.float32).max).all()


def test_exact_ties_lower_id_first(engine):
    rng = np.random.default_rng(2)
    base = _rand_unit(rng, 50, 384)
    X = np.concatenate([base, base[::-1], base[:10]], axis=0)  # every row duplicated at least once
    Q = base[:4] + 0.01 * _rand_unit(rng, 4, 384)
    for metric in (0, 1):
        idx = _index(384, metric)
        idx.add(X)
        D, I = idx.search(Q, 20)
        _, Ir, D64 = od.flat_search(X, Q, 20, metric)
        # duplicates produce bit-equal scores on the GPU too (same row bytes, same order),
        # so the id order inside a tie group is fully determined: ascending id.
        for qi in range(4):
            for j in range(19):
                if D[qi, j] == D[qi, j + 1]:
                    assert I[qi, j] < I[qi, j + 1]
        od.check_topk(D, I, D64, Ir, RTOL, ATOL, _scorer(X, Q, metric))
        # the duplicate pairs must both be present
        assert set(I[0, :2].tolist()) <= set(np.flatnonzero((X == X[I[0, 0]]).all(axis=1)).tolist())


def test_all_equal_scores(engine):
    X = np.ones((1000, 8), np.float32)
    Q = np.ones((1, 8), np.float32)
    idx = _index(8)
    idx.add(X)
    D, I = idx.search(Q, 10)
    assert I[0].tolist() == list(range(10)) and (D == 8.0).all()


def test_row_at_a_time_add_and_interleaved_search(engine):
    # the reference adds one row per call (vector_storage.py:105)
    rng = np.random.default_rng(3)
    X = _rand_unit(rng, 300, 384)
    Q = _rand_unit(rng, 1, 384)
    idx = _index(384)
    for i in range(150):
        idx.add(X[i:i + 1])
    _check(idx, X[:150], Q, 10, 0)
    for i in range(150, 300):
        idx.add(X[i:i + 1])
    assert idx.ntotal == 300
    _check(idx, X, Q, 10, 0)
    idx.reset()
    assert idx.ntotal == 0
    idx.add(X[:3])
    _check(idx, X[:3], Q, 10, 0)


def test_reconstruct_and_index_file_roundtrip(engine, tmp_path):
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(4)
    for d, cls in ((384, faiss.IndexFlatIP), (37, faiss.IndexFlatL2)):
        X = rng.standard_normal((777, d)).astype(np.float32)
        idx = cls(d)
        idx.add(X)
        assert idx.reconstruct_n().tobytes() == X.tobytes()
        p = str(tmp_path / ("index_%d.faiss" % d))
        faiss.write_index(idx, p)
        assert os.path.getsize(p) == 45 + X.nbytes
        idx2 = faiss.read_index(p)
        assert idx2.ntotal == 777 and idx2.d == d and idx2.metric_type == idx.metric_type
        assert idx2.reconstruct_n().tobytes() == X.tobytes()
        Q = rng.standard_normal((2, d)).astype(np.float32)
        D1, I1 = idx.search(Q, 5)
        D2, I2 = idx2.search(Q, 5)
        assert np.array_equal(I1, I2) and np.array_equal(D1, D2)


def test_results_independent_of_launch_shape(engine):
    # per-row summation order is fixed, so grid size / rows per tile / nt loads must not change a bit
    from rag_system_amd import _lib
    lib = _lib.engine()
    rng = np.random.default_rng(6)
    X = _rand_unit(rng, 20_011, 384)
    Q = _rand_unit(rng, 3, 384)
    idx = _index(384)
    idx.add(X)
    _lib.check(lib.rag_set_option(b"mfma_min_nq", 1000))   # the scan kernel, whose knobs these are
    D0, I0 = idx.search(Q, 30)
    try:
        for name, val in [("scan_blocks", 3), ("scan_blocks", 4096), ("scan_nt", 0), ("scan_rows", 2), ("scan_rows", 8)]:
            _lib.check(lib.rag_set_option(name.encode(), val))
            D1, I1 = idx.search(Q, 30)
            assert np.array_equal(I0, I1), name
            if name != "scan_rows":  # a different R changes the reduction tree -> last-bit changes allowed
                assert np.array_equal(D0, D1), name
            else:
                assert np.allclose(D0, D1, rtol=1e-6, atol=1e-7)
            _lib.check(lib.rag_set_option(b"scan_blocks", 0))
            _lib.check(lib.rag_set_option(b"scan_nt", 1))
            _lib.check(lib.rag_set_option(b"scan_rows", 0))
    finally:
        lib.rag_set_option(b"scan_blocks", 0)
        lib.rag_set_option(b"scan_nt", 1)
        lib.rag_set_option(b"scan_rows", 0)
        lib.rag_set_option(b"mfma_min_nq", 0)


def test_sharded_search_and_device_merge(engine):
    """Row-sharded corpus: per-shard search_dev with id_base + rag_topk_merge_dev == one index."""
    import torch
    from rag_system_amd import _lib
    lib = _lib.engine()
    rng = np.random.default_rng(8)
    n, d, nq, k = 30_000, 384, 5, 30
    X = _rand_unit(rng, n, d)
    X[20_000:20_010] = X[5:15]  # exact duplicates across shards -> tie order crosses shard borders
    Q = _rand_unit(rng, nq, d)
    for metric in (0, 1):
        whole = _index(d, metric)
        whole.add(X)
        Dw, Iw = whole.search(Q, k)
        bounds = [0, 9_000, 9_001, 21_000, n]
        G = len(bounds) - 1
        dev = torch.device("cuda:0")
        qd = torch.from_numpy(Q).to(dev)
        Dl = torch.empty((G, nq, k), dtype=torch.float32, device=dev)
        Il = torch.empty((G, nq, k), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        shards = []
        for g in range(G):
            sh = _index(d, metric)
            sh.add(X[bounds[g]:bounds[g + 1]])
            shards.append(sh)
            sh._b.search_dev(nq, qd.data_ptr(), k, Dl[g].data_ptr(), Il[g].data_ptr(), bounds[g], st)
        Dm = torch.empty((nq, k), dtype=torch.float32, device=dev)
        Im = torch.empty((nq, k), dtype=torch.int64, device=dev)
        _lib.check(lib.rag_topk_merge_dev(metric, G, nq, k, C.c_void_p(Dl.data_ptr()), C.c_void_p(Il.data_ptr()),
                                          C.c_void_p(Dm.data_ptr()), C.c_void_p(Im.data_ptr()), C.c_void_p(st)))
        torch.cuda.synchronize()
        assert np.array_equal(Im.cpu().numpy(), Iw)
        assert np.array_equal(Dm.cpu().numpy(), Dw)
        # and the CPU restatement of the merge agrees bit for bit
        Dr, Ir = od.topk_merge(metric, Dl.cpu().numpy(), Il.cpu().numpy())
        assert np.array_equal(Ir, Iw) and np.array_equal(Dr, Dw)
        # packed layout of the sharded searcher (ids then scores in one buffer per rank, one
        # all-gather): the strided merge through the searcher's own views
        from rag_system_amd.sharded import ShardedFlatSearcher, _hip_merge
        m = nq * k
        words = m + (m + 1) // 2
        pack = torch.zeros((G, words), dtype=torch.int64, device=dev)
        I_all = pack[:, :m].view(G, nq, k)
        D_all = pack[:, m:].view(torch.float32)[:, :m].view(G, nq, k)
        I_all.copy_(Il)
        D_all.copy_(Dl)
        Dm.zero_()
        Im.zero_()
        _hip_merge(metric, D_all, I_all, Dm, Im, st)
        torch.cuda.synchronize()
        assert np.array_equal(Im.cpu().numpy(), Iw) and np.array_equal(Dm.cpu().numpy(), Dw)
        # single-rank searcher writes through the packed views too
        side = torch.cuda.Stream(dev)
        s1 = ShardedFlatSearcher(lambda q, kk, Dloc, Iloc, sh: whole._b.search_dev(nq, q.data_ptr(), kk, Dloc.data_ptr(),
                                                                                  Iloc.data_ptr(), 0, sh),
                                 metric, nq, k, dev, stream=side)
        D1, I1 = s1.search(qd)
        side.synchronize()      # the searcher's own stream, not the device
        assert np.array_equal(I1.cpu().numpy(), Iw) and np.array_equal(D1.cpu().numpy(), Dw)


def test_full_size_10M_x_384_against_streaming_oracle(engine):
    """BASELINE headline size.  The corpus is generated on the device (bit-identical to the
    oracle's generator, checked above); the oracle regenerates rows on the fly on the host."""
    n, d, k, seed = 10_000_000, 384, 10, 1234
    idx = _index(d)
    idx._b.add_synthetic(n, seed, 0)
    # planted exact duplicates of two corpus rows appended at the end (tie rule at scale)
    dup = od.synth_rows(seed, 0, 2, d)
    idx.add(dup)
    assert idx.ntotal == n + 2
    Q = od.synth_queries(4321, 2, d)
    Q = np.concatenate([Q, dup[:1]], axis=0)  # third query = row 0 itself: score 1.0 twice
    D, I = idx.search(Q, k)
    Dr, Ir, D64 = od.flat_search_synth(seed, 0, n, d, Q, k)
    # fold the two appended rows into the oracle result (same sequential float64 order)
    _, Ie, De64 = od.flat_search(dup, Q, k)
    Ie = np.where(Ie >= 0, Ie + n, -1)
    Dl = np.stack([D64.astype(np.float32), De64.astype(np.float32)])
    Il = np.stack([Ir, Ie])
    Dm, Im = od.topk_merge(0, Dl, Il)

    def scorer(qi, ids):
        rows = np.stack([od.synth_rows(seed, int(i), 1, d)[0] if i < n else dup[int(i) - n] for i in ids])
        return rows.astype(np.float64) @ Q[qi].astype(np.float64)
    od.check_topk(D, I, Dm.astype(np.float64), Im, RTOL, ATOL, scorer)
    assert I[2, 0] == 0 and I[2, 1] == n, "duplicate of row 0 must follow it (lower id first)"
    # the same three queries inside batches of every size class -- stream kernel (1, 8), 16x16
    # register-fed kernel (16, 64), its chunked passes (100), LDS tile kernel (300) -- at full size
    filler = od.synth_queries(999, 300, d)
    for nq in (1, 8, 16, 64, 100, 300):
        B = filler[:nq].copy()
        pos = [0] if nq == 1 else [0, nq // 2, nq - 1]
        for j, p in enumerate(pos):
            B[p] = Q[j]
        Db, Ib = idx.search(B, k)
        sel = np.asarray(pos)
        od.check_topk(Db[sel], Ib[sel], Dm[:len(pos)].astype(np.float64), Im[:len(pos)], RTOL, ATOL, scorer)
        assert (Ib >= 0).all() and (Ib < n + 2).all() and np.all(Db[:, :-1] >= Db[:, 1:])
    # k = 100: list mode (3 queries), filter mode of the stream kernel (6) and of the 16x16 kernel (20)
    k2 = 100
    _, Ir2, D642 = od.flat_search_synth(seed, 0, n, d, Q, k2)
    _, Ie2, De642 = od.flat_search(dup, Q, k2)
    Ie2 = np.where(Ie2 >= 0, Ie2 + n, -1)
    Dm2, Im2 = od.topk_merge(0, np.stack([D642.astype(np.float32), De642.astype(np.float32)]), np.stack([Ir2, Ie2]))
    for nq in (3, 6, 20):
        B = filler[:nq].copy()
        pos = [0, nq // 2, nq - 1]
        for j, p in enumerate(pos):
            B[p] = Q[j]
        Db, Ib = idx.search(B, k2)
        sel = np.asarray(pos)
        od.check_topk(Db[sel], Ib[sel], Dm2.astype(np.float64), Im2, RTOL, ATOL, scorer)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d", [(1, 384), (300, 384), (4099, 768), (3001, 37), (20_011, 384), (1000, 20)])
def test_batched_mfma_search_matches_oracle(engine, n, d, metric):
    """nq >= 5 goes through the f32 MFMA tile kernel (flat_mfma_kernel.h) for k <= 128."""
    rng = np.random.default_rng(n * 7 + d + metric)
    X = _rand_unit(rng, n, d)
    if n > 100:
        X[n // 2] = X[3]          # exact duplicates inside and across row tiles
        X[n - 1] = X[3]
    idx = _index(d, metric)
    idx.add(X)
    for nq, k in [(5, 10), (19, 1), (33, 32), (64, 10), (100, 33), (256, 10), (300, 64), (1024, 10), (7, 90), (40, 128)]:
        Q = _rand_unit(rng, nq, d)
        Q[0] = X[min(3, n - 1)]   # a query equal to a (duplicated) row
        D, I = idx.search(Q, k)
        _, Ir, D64 = od.flat_search(X, Q, k, metric)
        # L2 in the MFMA path is ||x||^2 + ||q||^2 - 2<x,q> (what faiss does for nq >= 20):
        # absolute error ~1e-7 * (||x||^2 + ||q||^2), so near-zero distances need an absolute floor
        atol = ATOL if metric == 0 else 2e-6
        od.check_topk(D, I, D64, Ir, RTOL, atol, _scorer(X, Q, metric))


def test_batched_results_identical_to_single_queries_ids(engine):
    rng = np.random.default_rng(77)
    X = _rand_unit(rng, 30_000, 384)
    Q = _rand_unit(rng, 64, 384)
    idx = _index(384)
    idx.add(X)
    Db, Ib = idx.search(Q, 10)
    for qi in (0, 17, 63):
        D1, I1 = idx.search(Q[qi:qi + 1], 10)
        assert np.array_equal(I1[0], Ib[qi])
        assert np.allclose(D1[0], Db[qi], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("metric", [0, 1])
def test_k_above_256_multi_pass(engine, metric):
    """k > RAG_MAX_K is served in passes bounded by the previous pass' last entry."""
    rng = np.random.default_rng(91 + metric)
    X = _rand_unit(rng, 3000, 64)
    X[1000:1300] = X[0:300]            # 300 exact duplicates: pass borders fall inside tie groups
    idx = _index(64, metric)
    idx.add(X)
    for nq, k in [(1, 257), (3, 300), (2, 700), (6, 513), (1, 3000), (1, 4000)]:
        Q = _rand_unit(rng, nq, 64)
        D, I = idx.search(Q, k)
        _, Ir, D64 = od.flat_search(X, Q, k, metric)
        od.check_topk(D, I, D64, Ir, RTOL, ATOL, _scorer(X, Q, metric))
        for qi in range(nq):            # no id may appear twice across passes
            ids = I[qi][I[qi] >= 0]
            assert len(set(ids.tolist())) == len(ids)


def test_every_mfma_block_shape_and_ring_depth(engine):
    """All built MFMA block shapes / K steps / LDS ring depths must give the same ids and
    (bit-)identical scores: per-row summation order does not depend on the shape."""
    from rag_system_amd import _lib
    lib = _lib.engine()
    rng = np.random.default_rng(123)
    X = _rand_unit(rng, 9_001, 200)       # d = 200: K tail for both BK = 16 and BK = 32
    X[4500] = X[17]
    Q = _rand_unit(rng, 70, 200)
    idx = _index(200)
    idx.add(X)
    _, Ir, D64 = od.flat_search(X, Q, 16, 0)
    base = None
    try:
        for shape in range(1, 10):
            for stages in (2, 3):
                _lib.check(lib.rag_set_option(b"mfma_shape", shape))
                _lib.check(lib.rag_set_option(b"mfma_stages", stages))
                D, I = idx.search(Q, 16)
                od.check_topk(D, I, D64, Ir, RTOL, ATOL, _scorer(X, Q, 0))
                if base is None:
                    base = (D.copy(), I.copy())
                assert np.array_equal(I, base[1]), (shape, stages)
                assert np.array_equal(D, base[0]), (shape, stages)
    finally:
        lib.rag_set_option(b"mfma_shape", 0)
        lib.rag_set_option(b"mfma_stages", 0)


def _with_options(lib, opts):
    from rag_system_amd import _lib
    for name, val in opts.items():
        _lib.check(lib.rag_set_option(name.encode(), val))


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_mfma_filter_mode_matches_oracle(engine, storage):
    """Filter mode of the batch kernel (sample pass -> thresholds -> append -> select, with the
    list-mode redo for queries whose count falls outside [k, cap]) forced on small inputs."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(4242)
    rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
    try:
        _with_options(lib, {"mfma_filter_k": 1, "mfma_filter_min_rows": 1})
        for metric in (0, 1):
            for n, d in [(300, 64), (20_011, 384), (70_001, 72)]:
                X = _rand_unit(rng, n, d)
                X[n // 2] = X[3]
                X[n - 1] = X[3]
                idx = faiss.IndexFlat(d, metric, storage=storage)
                idx.add(X)
                Xr = rnd(X)
                for nq, k in [(5, 10), (33, 1), (64, 100), (130, 50), (300, 128)]:
                    Q = _rand_unit(rng, nq, d)
                    Q[0] = X[3]
                    Qr = rnd(Q)
                    D, I = idx.search(Q, k)
                    _, Ir, D64 = od.flat_search(Xr, Qr, k, metric)
                    od.check_topk(D, I, D64, Ir, RTOL, ATOL if metric == 0 else 4e-6, _scorer(Xr, Qr, metric))
        # adversarial row orders for the 1-in-32 tile sample (row tiles of 256 rows)
        n, d, k = 40_000, 32, 20
        Q = _rand_unit(rng, 12, d)
        base = _rand_unit(rng, n, d) * 0.01
        tile = np.arange(n) // 256
        cases = {}
        a = base.copy(); a[:] = a[0]                                  # every score equal: overflow -> redo
        cases["all_equal"] = a
        b = base.copy(); b[(tile % 32 == 0)] *= 100.0                 # the best rows are all in sampled tiles: count < k... or exact
        cases["best_in_sample"] = b
        c = base.copy(); c[(tile % 32 == 5)] *= 100.0                 # the best rows are never sampled: huge counts
        cases["best_outside_sample"] = c
        e = base.copy(); e[:300] = Q[0] * 3.0                         # 300 exact ties at the top for query 0
        cases["ties_at_top"] = e
        for name, X in cases.items():
            idx = faiss.IndexFlat(d, 0, storage=storage)
            idx.add(X)
            Xr, Qr = rnd(X), rnd(Q)
            D, I = idx.search(Q, k)
            _, Ir, D64 = od.flat_search(Xr, Qr, k, 0)
            od.check_topk(D, I, D64, Ir, RTOL, ATOL, _scorer(Xr, Qr, 0))
            if name in ("all_equal", "ties_at_top"):
                assert np.array_equal(I[0], Ir[0]), name          # exact ties: lower id first
    finally:
        _with_options(lib, {"mfma_filter_k": 0, "mfma_filter_min_rows": 1 << 19})


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_pipelined_filter_kernel_equals_tile_kernel(engine, storage):
    """Block shape 10 (flat_mfma_pipe.h: software-pipelined K loop, 3- and 4-stage ring) against
    the oracle and bit-for-bit against shape 4 (same K order per accumulator), both metrics, with
    partial last row tiles, fewer queries than a block's 256 columns, and several query tiles."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(777)
    rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
    try:
        _with_options(lib, {"mfma_filter_k": 1, "mfma_filter_min_rows": 1, "wide_max_nq": -1})
        for metric in (0, 1):
            for n, d in [(70_001, 64), (33_000, 384)]:   # rows of 2 / 12 (bf16) or 4 / 24 (fp32) K stages
                X = _rand_unit(rng, n, d)
                X[n // 2] = X[3]
                X[n - 1] = X[3]
                idx = faiss.IndexFlat(d, metric, storage=storage)
                idx.add(X)
                Xr = rnd(X)
                for nq, k in [(130, 50), (300, 100), (600, 10)]:
                    Q = _rand_unit(rng, nq, d)
                    Q[0] = X[3]
                    Qr = rnd(Q)
                    _with_options(lib, {"mfma_shape": 4, "mfma_stages": 0})
                    D0, I0 = idx.search(Q, k)
                    assert "tile filter" in idx._b.last_kernel(), idx._b.last_kernel()
                    _, Ir, D64 = od.flat_search(Xr, Qr, k, metric)
                    for stages in (0, 3):
                        _with_options(lib, {"mfma_shape": 10, "mfma_stages": stages})
                        D, I = idx.search(Q, k)
                        od.check_topk(D, I, D64, Ir, RTOL, ATOL if metric == 0 else 4e-6, _scorer(Xr, Qr, metric))
                        assert np.array_equal(I, I0) and np.array_equal(D, D0), (metric, n, d, nq, k, stages)
    finally:
        _with_options(lib, {"mfma_filter_k": 0, "mfma_filter_min_rows": 1 << 19, "wide_max_nq": 0, "mfma_shape": 0,
                            "mfma_stages": 0})


def test_mfma_filter_mode_default_thresholds(engine):
    """600k x 64 rows, k = 50: filter mode by default; identical ids to list mode."""
    from rag_system_amd import _lib
    lib = _lib.engine()
    rng = np.random.default_rng(99)
    X = _rand_unit(rng, 600_000, 64)
    Q = _rand_unit(rng, 40, 64)
    idx = _index(64)
    idx.add(X)
    D, I = idx.search(Q, 50)
    _, Ir, D64 = od.flat_search(X, Q, 50, 0)
    od.check_topk(D, I, D64, Ir, RTOL, ATOL, _scorer(X, Q, 0))
    try:
        _with_options(lib, {"mfma_filter_k": 1000})
        D2, I2 = idx.search(Q, 50)
    finally:
        _with_options(lib, {"mfma_filter_k": 0})
    # 40 queries: the default pass is the 16x16 register-fed kernel (flat_wide.hip), the forced
    # list mode the LDS tile kernel -> same ids, scores equal up to the summation order
    assert np.array_equal(I, I2) and np.allclose(D, D2, rtol=2e-6, atol=2e-7)
    try:
        _with_options(lib, {"wide_max_nq": -1})          # tile kernel in filter mode: bit-identical to its list mode
        D3, I3 = idx.search(Q, 50)
    finally:
        _with_options(lib, {"wide_max_nq": 0})
    assert np.array_equal(I2, I3) and np.array_equal(D2, D3)


@pytest.mark.parametrize("metric", [0, 1])
def test_large_batches_take_the_pipelined_filter_flow_at_every_k(engine, metric):
    """More than 128 queries over rows of whole 128-byte stages on a shard of 512k+ rows: the default route is
    the pipelined tile kernel's filter flow whatever k is (r03; fp32 rows here, 600k x 64, k = 5 and 1);
    ids as the oracle's, and as the list-mode tile kernel's."""
    from rag_system_amd import _lib
    lib = _lib.engine()
    rng = np.random.default_rng(2024 + metric)
    X = _rand_unit(rng, 600_000, 64)
    X[599_999] = X[7]
    X[300_000] = X[7]
    Q = _rand_unit(rng, 200, 64)
    Q[0] = X[7]
    idx = _index(64, metric)
    idx.add(X)
    for k in (5, 1):
        D, I = idx.search(Q, k)
        assert "tile filter" in idx._b.last_kernel(), idx._b.last_kernel()
        _, Ir, D64 = od.flat_search(X, Q, k, metric)
        od.check_topk(D, I, D64, Ir, RTOL, ATOL if metric == 0 else 4e-6, _scorer(X, Q, metric))
        try:
            _with_options(lib, {"mfma_filter_k": 1000})
            D2, I2 = idx.search(Q, k)
            assert "tile list" in idx._b.last_kernel(), idx._b.last_kernel()
        finally:
            _with_options(lib, {"mfma_filter_k": 0})
        assert np.array_equal(I, I2) and np.array_equal(D, D2)   # same K order per accumulator: bit-identical
