"""ctypes binding of oracle/flat_ref.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates faiss IndexFlatIP / IndexFlatL2 .search as called at vector_storage.py:196.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

METRIC_IP = 0
METRIC_L2 = 1

_lib = None


def build(force=False):
    src = os.path.join(HERE, "flat_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp, i64, u64 = C.c_void_p, C.c_int64, C.c_uint64
        L.synth_rows.argtypes = [u64, i64, i64, C.c_int, vp]
        L.synth_rows.restype = None
        L.flat_search_ref.argtypes = [vp, i64, C.c_int, vp, i64, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.flat_search_ref.restype = None
        L.flat_search_synth_ref.argtypes = [u64, i64, i64, C.c_int, vp, i64, C.c_int, C.c_int, vp, vp, vp]
        L.flat_search_synth_ref.restype = None
        L.flat_search_synth_ref2.argtypes = [u64, i64, i64, C.c_int, vp, i64, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.flat_search_synth_ref2.restype = None
        L.bf16_round_array.argtypes = [vp, i64, vp]
        L.bf16_round_array.restype = None
        L.topk_merge_ref.argtypes = [C.c_int, C.c_int, i64, C.c_int, vp, vp, vp, vp]
        L.topk_merge_ref.restype = None
        L.flat_search_cpu_baseline.argtypes = [vp, i64, C.c_int, vp, i64, C.c_int, C.c_int, vp, vp]
        L.flat_search_cpu_baseline.restype = None
        L.oracle_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def synth_rows(seed, first_global_row, n, d):
    out = np.empty((n, d), dtype=np.float32)
    lib().synth_rows(seed, first_global_row, n, d, _p(out))
    return out


def synth_queries(seed, nq, d):
    """Queries are rows of the same generator under a different seed."""
    return synth_rows(seed, 0, nq, d)


def flat_search(X, Q, k, metric=METRIC_IP, accum64=True):
    """-> (D float32 (nq,k), I int64 (nq,k), D64 float64 (nq,k))."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    if X.ndim != 2:
        X = X.reshape(-1, Q.shape[-1])
    Q = Q.reshape(-1, X.shape[1]) if X.shape[0] else Q.reshape(-1, Q.shape[-1])
    n, d = X.shape[0], Q.shape[1]
    nq = Q.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    D64 = np.empty((nq, k), dtype=np.float64)
    lib().flat_search_ref(_p(X), n, d, _p(Q), nq, k, metric, 1 if accum64 else 0, _p(D), _p(I), _p(D64))
    return D, I, D64


def bf16_round(a):
    """float32 -> bfloat16 -> float32 (round-to-nearest-even): the storage rounding of a
    RAG_DTYPE_BF16 index, applied to rows AND queries."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    lib().bf16_round_array(_p(a), a.size, _p(out))
    return out


def flat_search_synth(seed, first_global_row, n, d, Q, k, metric=METRIC_IP, bf16=False):
    """bf16=True: rows are rounded to bfloat16 after generation; Q must already be rounded."""
    Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, d)
    nq = Q.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    D64 = np.empty((nq, k), dtype=np.float64)
    lib().flat_search_synth_ref2(seed, first_global_row, n, d, _p(Q), nq, k, metric, 1 if bf16 else 0,
                                 _p(D), _p(I), _p(D64))
    return D, I, D64


def topk_merge(metric, Dl, Il):
    """Dl, Il: [nlists, nq, k] -> merged (D, I) [nq, k]."""
    Dl = np.ascontiguousarray(Dl, dtype=np.float32)
    Il = np.ascontiguousarray(Il, dtype=np.int64)
    nlists, nq, k = Dl.shape
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().topk_merge_ref(metric, nlists, nq, k, _p(Dl), _p(Il), _p(D), _p(I))
    return D, I


def cpu_baseline_search(X, Q, k, nthreads=1):
    """The timed CPU leg of bench.py: faiss-cpu-style fp32 flat IP scan."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, X.shape[1])
    nq = Q.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().flat_search_cpu_baseline(_p(X), X.shape[0], X.shape[1], _p(Q), nq, k, nthreads, _p(D), _p(I))
    return D, I


def num_threads():
    return lib().oracle_num_threads()


def check_topk(D, I, D_ref64, I_ref, rtol=1e-5, atol=1e-6, exact_scores=None):
    """Parity bar of BASELINE.json: scores within `rtol` relative (plus `atol` absolute for
    scores near zero), ids identical except where the scores involved tie within tolerance.

    exact_scores(qi, ids) -> float64 scores of arbitrary ids for query qi (used to validate
    an id that differs from the oracle's: it must score within tolerance of the oracle's
    entry at that rank).  Returns the number of positions whose id differed (all near-ties).
    """
    D = np.asarray(D, dtype=np.float64)
    D_ref64 = np.asarray(D_ref64, dtype=np.float64)
    assert D.shape == D_ref64.shape == I.shape == I_ref.shape, (D.shape, D_ref64.shape, I.shape, I_ref.shape)
    pad = I_ref < 0
    assert np.array_equal(I < 0, pad), "padding positions differ"
    assert np.array_equal(D[pad], D_ref64[pad]), "pad scores differ"
    tol = atol + rtol * np.abs(D_ref64)
    bad = (~pad) & (np.abs(D - D_ref64) > tol)
    assert not bad.any(), "scores differ beyond tolerance at %s: %s vs %s" % (
        np.argwhere(bad)[:5].tolist(), D[bad][:5], D_ref64[bad][:5])
    # every returned id at most once per query (a repeated id could otherwise pass as a near-tie of two ranks)
    for qi in range(I.shape[0]):
        vals, counts = np.unique(I[qi][~pad[qi]], return_counts=True)
        assert (counts == 1).all(), "query %d: ids returned more than once: %s" % (qi, vals[counts > 1][:6].tolist())
    diff = (~pad) & (I != I_ref)
    ndiff = int(diff.sum())
    if ndiff:
        for qi, j in np.argwhere(diff):
            # a differing id is only acceptable as a near-tie: its true score must be within
            # tolerance of the oracle's score at this rank
            if exact_scores is not None:
                s_true = float(exact_scores(int(qi), np.array([I[qi, j]]))[0])
                assert abs(s_true - D_ref64[qi, j]) <= 2 * tol[qi, j], (
                    "id mismatch at q=%d rank=%d: got %d (true score %r) expected %d (score %r)"
                    % (qi, j, I[qi, j], s_true, I_ref[qi, j], D_ref64[qi, j]))
            else:
                # without a scorer: the id must appear in the oracle list within tolerance
                near = np.abs(D_ref64[qi] - D_ref64[qi, j]) <= 2 * tol[qi, j]
                assert I[qi, j] in I_ref[qi][near], (
                    "id mismatch at q=%d rank=%d: got %d expected %d and no near-tie explains it"
                    % (qi, j, I[qi, j], I_ref[qi, j]))
    return ndiff
