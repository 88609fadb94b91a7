"""GPU parity tests of the 16x16 register-fed batch kernel (csrc/flat_wide.hip): 17..64 queries,
sample pass (wave maxima -> threshold) + filter pass + exact selection, redo by the tile kernel.

Oracle: float64 restatement of the IndexFlat.search call at vector_storage.py:196 (oracle/dense.py); bar: 1e-5 relative, ids identical except near-ties (BASELINE.json).
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
def test_wide_kernel_matches_oracle(engine, metric, storage):
    """Forced on small shards (mfma_filter_min_rows = 1).  d = 37, 20, 72, 1000: rows that end
    inside a 128-byte step; n = 16, 17, 300: ragged and window-shifted last tiles."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    try:
        _opts(lib, mfma_filter_min_rows=1)
        for n, d in [(16, 64), (17, 384), (300, 384), (4099, 768), (3001, 37), (20_011, 384), (1000, 20),
                     (2000, 1000), (50_003, 72)]:
            rng = np.random.default_rng(n * 5 + d + metric)
            X = _rand_unit(rng, n, d)
            if n > 100:
                X[n // 2] = X[3]
                X[n - 1] = X[3]
            idx = faiss.IndexFlat(d, metric, storage=storage)
            idx.add(X)
            # 65..192 queries: two or three passes of at most 64 queries each
            for nq, k in [(17, 10), (32, 1), (33, 64), (48, 33), (49, 100), (64, 128), (64, 10), (20, 5),
                          (65, 10), (100, 33), (129, 5), (192, 100)]:
                Q = _rand_unit(rng, nq, d)
                Q[0] = X[min(3, n - 1)]
                _check(idx, X, Q, k, metric, storage)
    finally:
        _opts(lib, mfma_filter_min_rows=1 << 19)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_wide_kernel_small_batches_and_launch_shapes(engine, storage):
    """wide_min_nq = 1 sends small batches (one query tile) through the kernel too; the result
    does not depend on the waves per block; ids equal to the tile kernel's."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(23)
    X = _rand_unit(rng, 30_011, 200)
    idx = faiss.IndexFlat(200, 0, storage=storage)
    idx.add(X)
    try:
        _opts(lib, mfma_filter_min_rows=1, wide_min_nq=1)
        for nq, k in [(3, 10), (16, 50), (40, 100), (64, 20)]:
            Q = _rand_unit(rng, nq, 200)
            D0, I0 = _check(idx, X, Q, k, 0, storage)
            for wpb in (1, 4, 12):
                _opts(lib, wide_wpb=wpb)
                D1, I1 = _check(idx, X, Q, k, 0, storage)
                # the block count changes the sample (one value per block), so a query may take the redo
                # path of the tile kernel: same ids, scores equal up to the summation order
                assert np.array_equal(I0, I1) and np.allclose(D0, D1, rtol=2e-6, atol=2e-7), (nq, k, wpb)
            _opts(lib, wide_wpb=0, wide_max_nq=-1, stream_max_nq=-1)
            D2, I2 = _check(idx, X, Q, k, 0, storage)
            _opts(lib, wide_max_nq=0, stream_max_nq=0)
            assert np.array_equal(I0, I2), (nq, k)
            assert np.allclose(D0, D2, rtol=2e-6, atol=2e-7)
    finally:
        _opts(lib, mfma_filter_min_rows=1 << 19, wide_min_nq=0, wide_wpb=0, wide_max_nq=0, stream_max_nq=0)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_wide_kernel_adversarial_row_orders(engine, storage):
    """Row orders that defeat the 1-in-32 sample of 16-row tiles (thresholds too high or too low,
    all scores equal, long tie runs at the top): the redo path must still give the exact result."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(778)
    rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
    try:
        _opts(lib, mfma_filter_min_rows=1)
        n, d, k = 40_000, 32, 20
        Q = _rand_unit(rng, 40, d)
        base = _rand_unit(rng, n, d) * 0.01
        wt = np.arange(n) // 16
        cases = {}
        a = base.copy(); a[:] = a[0]
        cases["all_equal"] = a
        b = base.copy(); b[(wt % 32 == 0)] *= 100.0
        cases["best_in_sample"] = b
        c = base.copy(); c[(wt % 32 == 5)] *= 100.0
        cases["best_outside_sample"] = c
        e = base.copy(); e[:300] = Q[0] * 3.0
        cases["ties_at_top"] = e
        f = base.copy(); f[:16] *= 1000.0                  # every top row in ONE sampled tile (one wave maximum)
        cases["best_in_one_tile"] = f
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
        _opts(lib, mfma_filter_min_rows=1 << 19)


def test_wide_kernel_default_on_a_large_shard(engine):
    """2M x 96 synthetic rows, 64 and 33 queries: taken by default; exact against the streaming
    oracle and identical ids to the tile kernel."""
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    n, d, seed = 2_000_000, 96, 99
    idx = faiss.IndexFlat(d, 0)
    idx._b.add_synthetic(n, seed, 0)
    Q = od.synth_queries(6, 64, d)
    Q[1] = od.synth_rows(seed, 1_234_567, 1, d)[0]
    Q = np.concatenate([Q, od.synth_queries(7, 36, d)])
    for nq, k in [(64, 10), (33, 100), (100, 10)]:
        D, I = idx.search(Q[:nq], k)
        _, Ir, D64 = od.flat_search_synth(seed, 0, n, d, Q[:nq], k)

        def scorer(qi, ids):
            rows = np.stack([od.synth_rows(seed, int(i), 1, d)[0] for i in ids])
            return rows.astype(np.float64) @ Q[qi].astype(np.float64)
        od.check_topk(D, I, D64, Ir, RTOL, ATOL, scorer)
        assert I[1, 0] == 1_234_567
        try:
            _opts(lib, wide_max_nq=-1)
            D2, I2 = idx.search(Q[:nq], k)
        finally:
            _opts(lib, wide_max_nq=0)
        assert np.array_equal(I, I2), (nq, k)
