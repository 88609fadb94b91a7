"""GPU parity tests of the opt-in bfloat16 prefilter of an fp32 index (rag_flat_prefilter, csrc/flat.hip).

The prefilter must never change the result: candidates come from a bfloat16 copy of the rows, scores from the
fp32 rows, and a per-query bound decides whether the candidates provably contain the top k.  Checked against
the float64 oracle (oracle/dense.py, restating the IndexFlat.search call of vector_storage.py:196) to the same
bar as the plain fp32 search, against that search itself, and with inputs built to defeat the bound (rows that
differ from each other by less than a bfloat16 rounding; one row a thousand times longer than the rest; too
few candidates), where the device-side redo on the fp32 rows has to deliver the result.
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


def _check(idx, X, Q, k, metric, rtol=RTOL):
    D, I = idx.search(Q, k)
    _, Ir, D64 = od.flat_search(X, Q, k, metric)
    od.check_topk(D, I, D64, Ir, rtol, ATOL if metric == 0 else 4e-6, _scorer(X, Q, metric))
    return D, I


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d", [(40_000, 384), (33_001, 100), (70_003, 768), (50_000, 37)])
def test_prefilter_equals_plain_search_and_oracle(engine, n, d, metric):
    """Every batch size class (stream kernel 1-8, wide 9-64, chunks 65-192, tile kernel above) and k up to 26,
    duplicated rows included (exact ties inside the candidates)."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(n + d + metric)
    X = _rand_unit(rng, n, d)
    X[n // 2] = X[3]
    X[n - 1] = X[3]
    idx = faiss.IndexFlat(d, metric)
    idx.add(X)
    for nq, k in [(1, 10), (1, 1), (2, 26), (8, 10), (5, 16), (9, 10), (33, 26), (64, 5), (100, 10), (300, 20)]:
        Q = _rand_unit(rng, nq, d)
        Q[0] = X[3]
        idx.set_prefilter(False)
        D0, I0 = _check(idx, X, Q, k, metric)
        assert "prefilter" not in idx._b.last_kernel()
        idx.set_prefilter(True)
        D1, I1 = _check(idx, X, Q, k, metric)
        assert idx._b.last_kernel().startswith("prefilter"), idx._b.last_kernel()
        redone = idx._b.prefilter_redone()
        assert redone <= 2, (nq, k, redone)   # the duplicated rows aside, random unit rows are proved at once
        same = I0 == I1
        # different ids only where the two fp32 scores tie to the bar
        assert np.allclose(D0[~same], D1[~same], rtol=RTOL, atol=4e-6), (nq, k)
        assert same.mean() > 0.99
        assert np.allclose(D0, D1, rtol=RTOL, atol=4e-6)
    # k > 26: served without the prefilter
    idx.search(_rand_unit(rng, 2, d), 27)
    assert "prefilter" not in idx._b.last_kernel()


@pytest.mark.parametrize("metric", [0, 1])
def test_prefilter_rows_closer_than_a_bf16_rounding(engine, metric):
    """20 000 rows inside a ball of relative radius 1e-3 around one direction (all of them round to nearly
    the same bfloat16 row) plus 30 000 ordinary rows: the candidates of the copy are an arbitrary subset of the
    cluster, the bound cannot close, and the fp32 redo has to return the true order."""
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(5 + metric)
    d = 128
    base = _rand_unit(rng, 1, d)
    cluster = base + 1e-3 * rng.standard_normal((20_000, d)).astype(np.float32) / np.sqrt(d)
    X = np.concatenate([_rand_unit(rng, 30_000, d), cluster.astype(np.float32)])
    X = X[rng.permutation(len(X))]
    idx = faiss.IndexFlat(d, metric)
    idx.add(X)
    idx.set_prefilter(True)
    for nq, k in [(1, 10), (4, 20), (40, 10), (200, 10)]:
        Q = _rand_unit(rng, nq, d)
        Q[0] = base[0]
        Q[-1] = base[0] + 0.01 * Q[-1]
        _check(idx, X, Q, k, metric)
        assert idx._b.last_kernel().startswith("prefilter")
        assert idx._b.prefilter_redone() >= min(2, nq)     # at least the queries aimed at the cluster


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("side", [-1, 1])
def test_prefilter_components_next_to_a_bf16_rounding_midpoint(engine, metric, side):
    """Every component of the query and of half the rows sits just below (side = -1) or just above (+1) a
    bfloat16 rounding midpoint, so ALL roundings go the same way: ||bf16(q)||^2 differs from ||q||^2 by the
    full 2^-7 relative, the inner products of the copy are biased by the whole bound, and the cluster rows
    (parallel to the query, closer together than a rounding) are indistinguishable in the copy.  The shadow's
    L2 key holds the fp32 norms of both operands (sync_shadow, prepare_l2_norms), so the outside-row bound must
    NOT be shifted by ||bf16(q)||^2 - ||q||^2 (with the shift the proof could close on a wrong
    candidate set).  Ids must be those of the plain fp32 search."""
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(900 + metric + side)
    d, n_cl, n_bg = 64, 20_000, 20_000
    # bfloat16 keeps 8 significant bits: m * 2^e with m in [128, 256); midpoint of (m, m+1) = m + 0.5
    m = rng.integers(128, 140, size=d).astype(np.float64)
    qv = ((m + 0.5 + side * 2.0 ** -9) / 128.0 / np.sqrt(d)).astype(np.float32)
    lens = 1.0 + 2e-4 * rng.standard_normal(n_cl)           # same direction, lengths inside one rounding step
    cluster = (qv[None, :].astype(np.float64) * lens[:, None]).astype(np.float32)
    X = np.concatenate([cluster, _rand_unit(rng, n_bg, d) * np.float32(np.linalg.norm(qv))])
    X = X[rng.permutation(len(X))]
    plain = faiss.IndexFlat(d, metric)
    plain.add(X)
    idx = faiss.IndexFlat(d, metric)
    idx.add(X)
    idx.set_prefilter(True)
    for nq, k in [(1, 10), (3, 16), (20, 10), (100, 10)]:
        Q = np.repeat(qv[None, :], nq, axis=0).copy()
        Q[1:] *= (1.0 + 1e-4 * rng.standard_normal((nq - 1, 1))).astype(np.float32)
        D0, I0 = _check(plain, X, Q, k, metric)
        D1, I1 = _check(idx, X, Q, k, metric)
        assert idx._b.last_kernel().startswith("prefilter")
        same = I0 == I1
        assert np.allclose(D0[~same], D1[~same], rtol=RTOL, atol=4e-6), (nq, k, int((~same).sum()))
        assert np.allclose(D0, D1, rtol=RTOL, atol=4e-6)
        # IP: the longest rows win and the copy cannot tell them apart; L2: the rows nearest in length
        assert idx._b.prefilter_redone() >= 1, (nq, k)


@pytest.mark.parametrize("metric", [0, 1])
def test_prefilter_one_long_row_and_nan(engine, metric):
    """max ||x|| enters the bound: one row 1000 times longer than the rest makes it useless (every query is
    redone), a NaN row makes it undefined (same).  Results still exact."""
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(77 + metric)
    d = 64
    X = _rand_unit(rng, 40_000, d)
    X[123] *= 1000.0
    idx = faiss.IndexFlat(d, metric)
    idx.add(X)
    idx.set_prefilter(True)
    Q = _rand_unit(rng, 5, d)
    _check(idx, X, Q, 10, metric)
    assert idx._b.prefilter_redone() == 5
    Q = _rand_unit(rng, 70, d)
    _check(idx, X, Q, 10, metric)
    assert idx._b.prefilter_redone() == 70
    # rows added later reach the copy too
    X2 = _rand_unit(rng, 3_000, d)
    X2[7] = Q[0]
    idx.add(X2)
    Xall = np.concatenate([X, X2])
    D, I = _check(idx, Xall, Q, 10, metric)
    assert I[0, 0] == 40_007 or metric == 0
    # reset empties the copy as well
    idx.reset()
    Y = _rand_unit(rng, 35_000, d)
    idx.add(Y)
    _check(idx, Y, Q, 10, metric)
    assert idx._b.last_kernel().startswith("prefilter")


def test_prefilter_too_few_candidates_are_redone(engine):
    """prefilter_kp = k: the k candidates of the copy are the only ones, so the bound closes only when the
    k-th exact score clears the k-th approximate score by the rounding margin -- most queries are redone.
    Either way the result is the oracle's."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(99)
    d, n = 384, 60_000
    X = _rand_unit(rng, n, d)
    idx = faiss.IndexFlat(d, 0)
    idx.add(X)
    idx.set_prefilter(True)
    try:
        for kp, nq, k in [(10, 3, 10), (10, 50, 10), (16, 300, 16), (20, 7, 10)]:
            _opts(lib, prefilter_kp=kp)
            Q = _rand_unit(rng, nq, d)
            _check(idx, X, Q, k, 0)
            assert idx._b.last_kernel().startswith("prefilter kp%d" % kp)
            assert idx._b.prefilter_redone() >= (nq // 2 if kp == k else 0)
    finally:
        _opts(lib, prefilter_kp=0)


def test_prefilter_small_shards_and_bf16_index(engine):
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(3)
    X = _rand_unit(rng, 5_000, 96)
    idx = faiss.IndexFlat(96, 0)
    idx.add(X)
    idx.set_prefilter(True)
    Q = _rand_unit(rng, 3, 96)
    _check(idx, X, Q, 10, 0)
    assert "prefilter" not in idx._b.last_kernel()      # below prefilter_min_rows
    try:
        _opts(lib, prefilter_min_rows=1)
        for n in (5_000, 100, 30):                     # fewer rows than candidates: the copy returns them all
            idx.reset()
            idx.add(X[:n])
            _check(idx, X[:n], Q, 10, 0)
            assert idx._b.last_kernel().startswith("prefilter")
            D, I = idx.search(Q, 50)                    # k > 26: direct
            assert "prefilter" not in idx._b.last_kernel()
    finally:
        _opts(lib, prefilter_min_rows=32768)
    b = faiss.IndexFlat(96, 0, storage="bf16")
    with pytest.raises(Exception):
        b.set_prefilter(True)


def test_prefilter_full_size_10M_x_384(engine):
    """BASELINE headline size with the prefilter on: the planted queries against the streaming oracle
    (oracle/dense.py flat_search_synth regenerates the rows on the host), every batch-size class against the
    plain fp32 search of the same index, and no query redone (unit rows of dimension 384: the k-th and the
    64th / 104th score lie several rounding margins apart)."""
    from rag_system_amd import faiss_compat as faiss
    n, d, k, seed = 10_000_000, 384, 10, 1234
    idx = faiss.IndexFlat(d, 0)
    idx._b.add_synthetic(n, seed, 0)
    dup = od.synth_rows(seed, 0, 2, d)
    idx.add(dup)                      # exact duplicates of rows 0 and 1 at the end: ties inside the candidates
    Q = od.synth_queries(4321, 2, d)
    Q = np.concatenate([Q, dup[:1]], axis=0)
    Dr, Ir, D64 = od.flat_search_synth(seed, 0, n, d, Q, k)
    _, Ie, De64 = od.flat_search(dup, Q, k)
    Ie = np.where(Ie >= 0, Ie + n, -1)
    Dm, Im = od.topk_merge(0, np.stack([D64.astype(np.float32), De64.astype(np.float32)]), np.stack([Ir, Ie]))

    def scorer(qi, ids):
        rows = np.stack([od.synth_rows(seed, int(i), 1, d)[0] if i < n else dup[int(i) - n] for i in ids])
        return rows.astype(np.float64) @ Q[qi].astype(np.float64)
    filler = od.synth_queries(999, 1024, d)
    for nq in (1, 3, 8, 16, 64, 100, 300, 1024):
        B = filler[:nq].copy()
        pos = [0] if nq == 1 else [0, nq // 2, nq - 1]
        for j, p in enumerate(pos):
            B[p] = Q[j]
        idx.set_prefilter(False)
        D0, I0 = idx.search(B, k)
        idx.set_prefilter(True)
        D1, I1 = idx.search(B, k)
        assert idx._b.last_kernel().startswith("prefilter"), idx._b.last_kernel()
        assert idx._b.prefilter_redone() == 0, (nq, idx._b.prefilter_redone())
        sel = np.asarray(pos)
        od.check_topk(D1[sel], I1[sel], Dm[:len(pos)].astype(np.float64), Im[:len(pos)], RTOL, ATOL, scorer)
        same = I0 == I1
        assert np.allclose(D0, D1, rtol=RTOL, atol=ATOL), nq
        assert same.mean() > 0.999, (nq, same.mean())          # ids differ only between fp32 near-ties
    if nq >= 3:
        assert I1[nq - 1, 0] == 0 and I1[nq - 1, 1] == n, "duplicate of row 0 must follow it (lower id first)"


def test_vector_storage_dense_prefilter_same_hits(engine):
    """VectorStorage.set_dense_prefilter: the hits of search / search_batch are those of the plain store, also
    after rebuild-style re-creation of the index (the switch is kept)."""
    from rag_system_amd import _lib
    from rag_system_amd.vector_storage import VectorStorage
    lib = _lib.engine()
    rng = np.random.default_rng(21)
    d, n = 64, 3000
    X = _rand_unit(rng, n, d)
    vs = VectorStorage(d)
    for i in range(n):
        vs.add_vector(X[i], "doc%d" % (i // 7), "chunk %d" % i, source="s")
    Q = _rand_unit(rng, 5, d)
    plain = [vs.search(q, top_k=5) for q in Q]
    plain_b = vs.search_batch(Q, top_k=5)
    try:
        _opts(lib, prefilter_min_rows=1)
        vs.set_dense_prefilter(True)
        got = [vs.search(q, top_k=5) for q in Q]
        assert vs.index._b.last_kernel().startswith("prefilter")
        got_b = vs.search_batch(Q, top_k=5)
        tok = vs.search_begin(Q[0], top_k=5)
        got_be = vs.search_end(tok)
    finally:
        _opts(lib, prefilter_min_rows=32768)
    for a, b in zip(plain + plain_b + [plain[0]], got + got_b + [got_be]):
        assert [h["chunk"] for h in a] == [h["chunk"] for h in b]
        assert np.allclose([h["score"] for h in a], [h["score"] for h in b], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("metric", [0, 1])
def test_prefilter_device_entry_point_with_id_base(engine, metric):
    """rag_flat_search_dev on a caller's stream with a non-zero id_base (one shard of a row-sharded index): the
    prefiltered search stays asynchronous -- proof and redo run on the device -- and returns global ids, for
    queries the bound proves and for queries it cannot (a cluster of near-duplicates)."""
    import torch
    from rag_system_amd import faiss_compat as faiss
    rng = np.random.default_rng(123 + metric)
    d, base = 96, 5_000_000_000
    dup = _rand_unit(rng, 1, d)
    cluster = (dup + 1e-3 * rng.standard_normal((15_000, d)).astype(np.float32) / np.sqrt(d)).astype(np.float32)
    X = np.concatenate([_rand_unit(rng, 45_000, d), cluster])
    X = X[rng.permutation(len(X))]
    idx = faiss.IndexFlat(d, metric)
    idx.add(X)
    idx.set_prefilter(True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    for nq, k in [(1, 10), (6, 5), (50, 10), (260, 8)]:
        Q = _rand_unit(rng, nq, d)
        Q[nq // 2] = dup[0]                      # this one cannot be proved
        with torch.cuda.stream(stream):
            q_dev = torch.from_numpy(Q).to(dev)
            D_dev = torch.empty((nq, k), dtype=torch.float32, device=dev)
            I_dev = torch.empty((nq, k), dtype=torch.int64, device=dev)
        idx._b.search_dev(nq, q_dev.data_ptr(), k, D_dev.data_ptr(), I_dev.data_ptr(), base, stream.cuda_stream)
        stream.synchronize()
        assert idx._b.last_kernel().startswith("prefilter")
        assert 1 <= idx._b.prefilter_redone() < max(nq, 2)
        D, I = D_dev.cpu().numpy(), I_dev.cpu().numpy()
        assert (I >= base).all()
        _, Ir, D64 = od.flat_search(X, Q, k, metric)
        od.check_topk(D, I - base, D64, Ir, RTOL, ATOL if metric == 0 else 4e-6, _scorer(X, Q, metric))
