"""Candidate-set stages on the MI355X: Reciprocal Rank Fusion and MMR selection.

Reference: SearchEngine.reciprocal_rank_fusion (search_engine.py:352-359) and the numeric
part of SearchEngine.mmr_selection (search_engine.py:419-465).  One workgroup per call
(csrc/cand.hip); these are latency-bound stages on tens of candidates.
"""
import ctypes as C

import numpy as np

from . import _lib


def _hip_rrf(ids, lens, k_const):
    lib = _lib.engine()
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    total = int(lens.sum())
    out_ids = np.empty(max(total, 1), dtype=np.int64)
    out_scores = np.empty(max(total, 1), dtype=np.float64)
    n_out = C.c_int64()
    _lib.check(lib.rag_rrf(len(lens), _lib.ptr(ids), _lib.ptr(lens), float(k_const), _lib.ptr(out_ids),
                           _lib.ptr(out_scores), C.byref(n_out)), lib)
    return out_ids[:n_out.value], out_scores[:n_out.value]


def _hip_mmr(cand, q, lam, top_k, byte_len, max_bytes):
    lib = _lib.engine()
    cand = np.ascontiguousarray(cand, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
    byte_len = np.ascontiguousarray(byte_len, dtype=np.int64)
    n, d = cand.shape
    order = np.empty(max(n, 1), dtype=np.int32)
    n_out = C.c_int()
    _lib.check(lib.rag_mmr(n, d, _lib.ptr(cand), _lib.ptr(q), float(lam), int(top_k), _lib.ptr(byte_len),
                           -1 if max_bytes is None else int(max_bytes), _lib.ptr(order), C.byref(n_out)), lib)
    return order[:n_out.value].tolist()


def _hip_mmr_stored(index_backend, rows, q, lam, top_k, byte_len, max_bytes):
    lib = _lib.engine()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
    byte_len = np.ascontiguousarray(byte_len, dtype=np.int64)
    n = len(rows)
    order = np.empty(max(n, 1), dtype=np.int32)
    n_out = C.c_int()
    _lib.check(lib.rag_flat_mmr(index_backend._h, n, _lib.ptr(rows), _lib.ptr(q), float(lam), int(top_k),
                                _lib.ptr(byte_len), -1 if max_bytes is None else int(max_bytes), _lib.ptr(order),
                                C.byref(n_out)), lib)
    return order[:n_out.value].tolist()


_rrf_impl = _hip_rrf
_mmr_impl = _hip_mmr
_mmr_stored_impl = _hip_mmr_stored


def set_impl(rrf=None, mmr=None, mmr_stored=None):
    """Test seam for host-logic tests; None restores the HIP engine."""
    global _rrf_impl, _mmr_impl, _mmr_stored_impl
    _rrf_impl = rrf or _hip_rrf
    _mmr_impl = mmr or _hip_mmr
    _mmr_stored_impl = mmr_stored or _hip_mmr_stored


def reciprocal_rank_fusion(rankings, k=60):
    """rankings: sequences of hashable doc ids, each already in ranking order (best first).
    Returns {doc_id: fused score} in first-appearance order, like the reference's dict."""
    table, names, flat, lens = {}, [], [], []
    for ranking in rankings:
        lens.append(len(ranking))
        for doc_id in ranking:
            i = table.get(doc_id)
            if i is None:
                i = table[doc_id] = len(names)
                names.append(doc_id)
            flat.append(i)
    if not flat:
        return {}
    ids, scores = _rrf_impl(np.asarray(flat, dtype=np.int64), np.asarray(lens, dtype=np.int64), k)
    return {names[int(i)]: float(s) for i, s in zip(ids, scores)}


def mmr_select(cand_emb, query_vec, byte_lens, lambda_param=0.5, top_k=None, max_context_bytes=None):
    """Indices of the selected candidates in selection order (search_engine.py:430-467)."""
    cand_emb = np.asarray(cand_emb, dtype=np.float32)
    if cand_emb.ndim != 2 or cand_emb.shape[0] == 0:
        return []
    top_k = cand_emb.shape[0] if top_k is None else top_k
    return _mmr_impl(cand_emb, query_vec, lambda_param, top_k, byte_lens, max_context_bytes)


def mmr_select_stored(index, rows, query_vec, byte_lens, lambda_param=0.5, top_k=None, max_context_bytes=None):
    """mmr_select with the candidates' embeddings taken from the flat index itself (rag_flat_mmr):
    candidate i is row rows[i] of `index` (a faiss_compat.IndexFlat on the HIP engine).  No
    re-encoding, no upload of embeddings."""
    rows = np.asarray(rows, dtype=np.int64).reshape(-1)
    if len(rows) == 0:
        return []
    backend = getattr(index, "_b", index)
    top_k = len(rows) if top_k is None else top_k
    return _mmr_stored_impl(backend, rows, query_vec, lambda_param, top_k, byte_lens, max_context_bytes)
