"""GPU robustness tests of the dense index through the C ABI: concurrent callers on one handle
(the reference UI starts a thread per query / ingest with no lock, UI_manager.py:753,883),
argument validation, non-finite inputs, extreme shapes."""
import ctypes as C
import threading

import numpy as np
import pytest

from oracle import dense as od

pytestmark = pytest.mark.gpu


def _rand_unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    return x


def test_concurrent_add_and_search_on_one_handle(engine):
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(17)
    d = 96
    base = _rand_unit(rng, 5000, d)
    extra = _rand_unit(rng, 4000, d)
    Q = _rand_unit(rng, 6, d)
    idx = faiss.IndexFlatIP(d)
    idx.add(base)
    errors = []

    def adder(lo, hi):
        try:
            for i in range(lo, hi):
                idx.add(extra[i:i + 1])                  # the reference's one-row add_vector pattern
        except Exception as e:                           # pragma: no cover
            errors.append(e)

    def searcher():
        try:
            for _ in range(60):
                D, I = idx.search(Q, 10)
                n_now = idx.ntotal
                assert ((I >= 0) & (I < n_now + 4000)).all()
                assert (D[:, :-1] >= D[:, 1:]).all()
        except Exception as e:                           # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=adder, args=(0, 2000)), threading.Thread(target=adder, args=(2000, 4000)),
               threading.Thread(target=searcher), threading.Thread(target=searcher)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert idx.ntotal == 9000
    # the two adders interleaved: the index holds the same multiset of rows; results match the
    # oracle evaluated on the index's own row order
    X = idx.reconstruct_n(0, 9000)
    assert sorted(map(bytes, X[5000:])) == sorted(map(bytes, extra))
    D, I = idx.search(Q, 10)
    _, Ir, D64 = od.flat_search(X, Q, 10, 0)
    od.check_topk(D, I, D64, Ir, 1e-5, 2e-6, lambda qi, ids: X[ids].astype(np.float64) @ Q[qi].astype(np.float64))


def test_argument_validation_through_the_c_abi(engine):
    from rag_system_amd import _lib
    lib = _lib.engine()
    h = C.c_void_p()
    assert lib.rag_flat_create(0, 0, 0, C.byref(h)) == -1 and b"d=0" in lib.rag_last_error()
    assert lib.rag_flat_create(16, 7, 0, C.byref(h)) == -1
    assert lib.rag_flat_create(16, 0, 9, C.byref(h)) == -1
    _lib.check(lib.rag_flat_create(16, 0, 0, C.byref(h)), lib)
    x = np.zeros((4, 16), np.float32)
    q = np.zeros((1, 16), np.float32)
    D = np.zeros((1, 4), np.float32)
    I = np.zeros((1, 4), np.int64)
    assert lib.rag_flat_add(h, -1, _lib.ptr(x)) == -1
    assert lib.rag_flat_add(h, 4, None) == -1
    _lib.check(lib.rag_flat_add(h, 4, _lib.ptr(x)), lib)
    assert lib.rag_flat_search(h, 1, _lib.ptr(q), 0, _lib.ptr(D), _lib.ptr(I)) == -1
    assert lib.rag_flat_search(h, 1, _lib.ptr(q), 16385, _lib.ptr(D), _lib.ptr(I)) == -1
    assert lib.rag_flat_search(h, 1, None, 4, _lib.ptr(D), _lib.ptr(I)) == -1
    assert lib.rag_flat_search(None, 1, _lib.ptr(q), 4, _lib.ptr(D), _lib.ptr(I)) == -1
    assert lib.rag_flat_copy_to_host(h, 2, 3, _lib.ptr(x)) == -1           # beyond ntotal
    assert lib.rag_set_option(b"no_such_option", 1) == -1
    assert lib.rag_topk_merge_dev(0, 0, 1, 4, _lib.ptr(D), _lib.ptr(I), _lib.ptr(D), _lib.ptr(I), None) == -1
    _lib.check(lib.rag_flat_search(h, 0, None, 4, None, None), lib)        # nq = 0 is a no-op
    _lib.check(lib.rag_flat_search(h, 1, _lib.ptr(q), 4, _lib.ptr(D), _lib.ptr(I)), lib)
    assert I.tolist() == [[0, 1, 2, 3]]                                    # all-equal scores: id order
    lib.rag_flat_free(h)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_non_finite_rows_never_displace_finite_ones(engine, storage):
    """A NaN score compares false against everything, so such rows are never selected (faiss
    leaves this unspecified); +-inf scores order like any other value."""
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(23)
    n, d = 3000, 64
    X = _rand_unit(rng, n, d)
    X[10, 3] = np.nan
    X[11] = np.inf
    X[2000, 0] = -np.inf
    Q = np.abs(_rand_unit(rng, 12, d)) + 0.01          # positive queries: row 11 scores +inf, row 2000 -inf
    idx = faiss.IndexFlat(d, 0, storage=storage)
    idx.add(X)
    for nq in (1, 12):
        D, I = idx.search(Q[:nq], 10)
        assert not np.isnan(D).any()
        assert (I[:, 0] == 11).all() and np.isposinf(D[:, 0]).all()
        assert not (I == 10).any() and not (I == 2000).any()
        finite = np.ones(n, bool)
        finite[[10, 11, 2000]] = False
        rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
        Xf, Qr = rnd(X[finite]), rnd(Q[:nq])
        ids = np.flatnonzero(finite)
        _, Ir, D64 = od.flat_search(Xf, Qr, 9, 0)
        od.check_topk(D[:, 1:], I[:, 1:], D64, ids[Ir], 1e-5, 2e-6,
                      lambda qi, rows: rnd(X[rows]).astype(np.float64) @ Qr[qi].astype(np.float64))


def test_extreme_shapes(engine):
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(31)
    for d, n, nq, k in [(1, 500, 3, 7), (2, 64, 70, 64), (8192, 40, 2, 5), (3, 1, 9, 3), (4097, 33, 5, 33)]:
        X, Q = _rand_unit(rng, n, d), _rand_unit(rng, nq, d)   # the tolerance is stated for normalised rows
        for metric in (0, 1):
            idx = faiss.IndexFlat(d, metric)
            idx.add(X)
            D, I = idx.search(Q, k)
            _, Ir, D64 = od.flat_search(X, Q, k, metric)
            X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
            od.check_topk(D, I, D64, Ir, 1e-5, 2e-6,
                          lambda qi, ids: (X64[ids] @ Q64[qi]) if metric == 0 else ((X64[ids] - Q64[qi]) ** 2).sum(1))


@pytest.mark.parametrize("which", ["default", "side"])
def test_device_search_is_ordered_on_the_callers_stream(engine, which):
    """rag_flat_search_dev is asynchronous on the stream it is given -- 0 is HIP's null stream, i.e.
    torch's default stream -- so a copy enqueued behind it on the same stream sees its result with no
    device-wide synchronisation.  A different query every step; the copy of step i is checked after
    synchronising that stream only.  (A stale or racing result would show as another query's ids.)"""
    import torch
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(71)
    n, d, k, steps = 200_000, 128, 10, 24
    X = _rand_unit(rng, n, d)
    Q = _rand_unit(rng, steps, d)
    idx = faiss.IndexFlatIP(d)
    idx.add(X)
    idx.search(Q[:1], k)                      # flush the staged rows
    _, Ir, D64 = od.flat_search(X, Q, k, 0)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream() if which == "default" else torch.cuda.Stream(dev)
    if which == "default":
        assert stream.cuda_stream == 0
    X64 = X.astype(np.float64)
    with torch.cuda.stream(stream):
        qd = torch.from_numpy(Q).to(dev)
        Dd = torch.empty((1, k), dtype=torch.float32, device=dev)
        Id = torch.empty((1, k), dtype=torch.int64, device=dev)
        Dh = torch.empty((steps, k), dtype=torch.float32).pin_memory()
        Ih = torch.empty((steps, k), dtype=torch.int64).pin_memory()
        stream.synchronize()
        for i in range(steps):
            idx._b.search_dev(1, qd[i:i + 1].data_ptr(), k, Dd.data_ptr(), Id.data_ptr(), 0, stream.cuda_stream)
            Dh[i:i + 1].copy_(Dd, non_blocking=True)
            Ih[i:i + 1].copy_(Id, non_blocking=True)
        stream.synchronize()                  # this stream only
    od.check_topk(Dh.numpy(), Ih.numpy(), D64, Ir, 1e-5, 2e-6,
                  lambda qi, ids: X64[ids] @ Q[qi].astype(np.float64))


def test_two_streams_on_one_handle_do_not_overlap(engine):
    """A handle's workspaces are shared by its searches: a search on another stream first waits for
    the handle's previous stream (include/rag_mi355x.h).  Alternating streams, no host
    synchronisation in between, different queries and k; every result must be its own query's."""
    import torch
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(72)
    n, d = 300_000, 64
    X = _rand_unit(rng, n, d)
    idx = faiss.IndexFlatIP(d)
    idx.add(X)
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream()]
    jobs = []
    for i in range(18):
        nq, k = [(1, 10), (5, 30), (20, 10), (3, 100)][i % 4]
        Q = _rand_unit(rng, nq, d)
        jobs.append((Q, k, torch.from_numpy(Q).to(dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                     torch.empty((nq, k), dtype=torch.int64, device=dev)))
    idx.search(jobs[0][0], 10)
    torch.cuda.synchronize()
    for i, (Q, k, qd, Dd, Id) in enumerate(jobs):
        idx._b.search_dev(Q.shape[0], qd.data_ptr(), k, Dd.data_ptr(), Id.data_ptr(), 0,
                          streams[i % 3].cuda_stream)
    torch.cuda.synchronize()
    X64 = X.astype(np.float64)
    for Q, k, _qd, Dd, Id in jobs:
        _, Ir, D64 = od.flat_search(X, Q, k, 0)
        od.check_topk(Dd.cpu().numpy(), Id.cpu().numpy(), D64, Ir, 1e-5, 2e-6,
                      lambda qi, ids: X64[ids] @ Q[qi].astype(np.float64))


def test_search_begin_end_equals_search_and_overlaps_sparse_scoring(engine):
    """rag_flat_search_begin / _end: the same (D, I) as rag_flat_search, bit for bit, with a BM25 scoring
    on the library stream between the two halves; misuse is refused with a clear error."""
    from rag_system_amd import faiss_compat as faiss
    from rag_system_amd.bm25 import BM25Okapi
    rng = np.random.default_rng(31)
    n, d = 60_000, 96
    X = rng.standard_normal((n, d)).astype(np.float32)
    idx = faiss.IndexFlatIP(d)
    idx.add(X)
    indptr = np.arange(0, 20 * 5000 + 1, 20, dtype=np.int64)
    bm = BM25Okapi.from_term_ids(indptr, rng.integers(0, 300, 20 * 5000))
    ref_scores = bm.get_scores([1, 2, 3])
    for nq, k in [(1, 10), (1, 90), (3, 7), (40, 100), (1, 300)]:
        Q = rng.standard_normal((nq, d)).astype(np.float32)
        D0, I0 = idx.search(Q, k)
        tok = idx.search_begin(Q, k)
        assert tok[0] == "pending"
        s = bm.get_scores([1, 2, 3])                       # sparse scoring beside the dense pass
        D1, I1 = idx.search_end(tok)
        assert np.array_equal(I0, I1) and np.array_equal(D0, D1), (nq, k)
        assert np.array_equal(s, ref_scores)
    # a batch beyond the pinned staging buffer is searched at once
    Qb = rng.standard_normal((12_000, d)).astype(np.float32)
    tok = idx.search_begin(Qb, 10)
    assert tok[0] == "done"
    D1, I1 = idx.search_end(tok)
    D0, I0 = idx.search(Qb, 10)
    assert np.array_equal(I0, I1)
    # misuse
    Q = rng.standard_normal((1, d)).astype(np.float32)
    tok = idx.search_begin(Q, 5)
    with pytest.raises(ValueError):
        idx.search_begin(Q, 5)
    with pytest.raises(ValueError):
        idx.search(Q, 5)
    idx.add(X[:10])                                        # an add between the halves is ordered behind the search
    D1, I1 = idx.search_end(tok)
    assert I1.max() < n
    with pytest.raises(ValueError):
        idx._b.search_end((1, 5))
    D2, I2 = idx.search(Q, 5)
    assert np.array_equal(I1, I2) or I2.max() >= n         # the ten new rows may now appear


def test_concurrent_queries_through_the_split_search_paths(engine):
    """search_rrf / search_hybrid / search_multi_representation run the dense pass and the TF-IDF
    scoring through *_begin / *_end, and a handle takes one begun search.  The reference's UI starts one thread
    per query (UI_manager.py:883): two threads on ONE SearchEngine must both get the results a single thread
    gets (the second waits; nobody is refused), and a blocking index.search from a third thread waits too."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _data import zipf_queries
    from test_host_logic import _build_store
    from rag_system_amd.search_engine import SearchEngine
    mgr, store, em = _build_store(n_docs=300, seed=41, dim=96)
    eng = SearchEngine(store, em)
    queries = [" ".join(t) for t in zipf_queries(8, 400, seed=5)]
    methods = ("search_rrf", "search_hybrid", "search_multi_representation")
    expected = {(m, q): getattr(eng, m)(q, top_k=10) for m in methods for q in queries}
    errors, done = [], []

    def worker(tid):
        try:
            for rep in range(6):
                for i, q in enumerate(queries):
                    m = methods[(i + tid + rep) % len(methods)]
                    got = getattr(eng, m)(q, top_k=10)
                    assert got == expected[(m, q)], (tid, m, q)
            done.append(tid)
        except Exception as e:      # noqa: BLE001 -- reported below
            errors.append((tid, repr(e)))

    def plain_searcher():
        try:
            for q in queries * 6:
                hits = store.search(em.embed_query(q), 5)
                assert len(hits) == 5
            done.append("plain")
        except Exception as e:      # noqa: BLE001
            errors.append(("plain", repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)] + [threading.Thread(target=plain_searcher)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    assert sorted(map(str, done)) == ["0", "1", "2", "plain"]
    # a thread that begins twice without ending is refused by the library, not dead-locked
    q = np.asarray(em.embed_query(queries[0]), np.float32).reshape(1, -1)
    tok = store.index.search_begin(q, 5)
    with pytest.raises(ValueError, match="has not been ended"):
        store.index.search_begin(q, 5)
    store.index.search_end(tok)
    assert store.index.search(q, 5)[1].shape == (1, 5)


def test_flat_mmr_beside_adds_that_grow_the_arena(engine):
    """rag_flat_mmr gathers candidate rows from the index arena; a concurrent rag_flat_add may
    reallocate that arena (the reference's UI adds documents beside its query thread, UI_manager.py:753,883).
    The selection runs under the handle's mutex until its kernel is enqueued, so every call returns the
    selection of the (unchanged) candidate rows while the index grows 30-fold underneath."""
    from rag_system_amd import faiss_compat as faiss
    from rag_system_amd.candidates import mmr_select, mmr_select_stored
    rng = np.random.default_rng(8)
    d = 256
    X0 = _rand_unit(rng, 4_000, d)
    idx = faiss.IndexFlatIP(d)
    idx.add(X0)
    rows = rng.choice(4_000, 40, replace=False)
    q = _rand_unit(rng, 1, d)[0]
    blen = rng.integers(10, 500, 40)
    expected = mmr_select(X0[rows], q, blen, 0.6, 12, 3000)
    errors, stop = [], threading.Event()

    def adder():
        try:
            for _ in range(30):
                idx.add(_rand_unit(rng, 4_000, d))       # several of these outgrow the arena
                idx.search(q.reshape(1, -1), 1)            # flush: the rows reach HBM now
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
        finally:
            stop.set()

    t = threading.Thread(target=adder)
    t.start()
    n_calls = 0
    while not stop.is_set() or n_calls < 50:
        got = mmr_select_stored(idx, rows, q, blen, 0.6, 12, 3000)
        assert got == expected, (n_calls, got, expected)
        n_calls += 1
    t.join(120)
    assert not errors, errors
    assert idx.ntotal == 4_000 * 31
