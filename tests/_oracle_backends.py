"""CPU stand-ins for the HIP handles, built on the oracle.  TESTS ONLY: they let the host logic
(faiss_compat codec, VectorStorage, SearchEngine, candidate shortcuts) run in the GPU-less
container through the product's test seams (faiss_compat.set_backend_factory,
bm25.set_csc_factory, candidates.set_impl).  Nothing in rag_system_amd/ refers to this file."""
import contextlib

import numpy as np

from oracle import dense as od
from oracle import fusion as ofu


class OracleFlatBackend:
    def __init__(self, d, metric):
        self.d, self.metric = d, metric
        self._rows = np.zeros((0, d), np.float32)

    @property
    def ntotal(self):
        return len(self._rows)

    def add(self, x):
        self._rows = np.concatenate([self._rows, np.asarray(x, np.float32)], axis=0)

    def search(self, q, k):
        D, I, _ = od.flat_search(self._rows, q, k, self.metric, accum64=True)
        return D, I

    def reset(self):
        self._rows = np.zeros((0, self.d), np.float32)

    def to_host(self, first=0, n=None):
        n = self.ntotal - first if n is None else n
        return self._rows[first:first + n].copy()


class OracleFlatBackendF32(OracleFlatBackend):
    """Orders like the GPU: by the float32 score it returns, lower id first among equal float32 scores (the plain
    oracle backend orders by the float64 score, which can separate two rows whose float32 scores are equal; a
    row-sharded search merges float32 lists and cannot know)."""

    def search(self, q, k):
        D, I = super().search(q, k)
        for r in range(len(D)):
            pad = I[r] < 0
            key = -D[r] if self.metric == 0 else D[r]
            order = np.lexsort((I[r], key, pad))
            D[r], I[r] = D[r][order], I[r][order]
        return D, I


class OracleCsc:
    def __init__(self, n_terms, n_rows, indptr, rows, vals):
        self.n_terms, self.n_rows = n_terms, n_rows
        self.indptr = np.asarray(indptr, np.int64)
        self.rows = np.asarray(rows, np.int64)
        self.vals = np.asarray(vals, np.float64)
        self._last = None

    def bm25_set(self, idf, doc_len, avgdl, k1, b):
        self.idf, self.doc_len = np.asarray(idf, np.float64), np.asarray(doc_len)
        self.avgdl, self.k1, self.b = avgdl, k1, b

    def _topk(self, s, k):
        order = np.lexsort((np.arange(len(s)), -s))[:k]
        return s[order], order.astype(np.int64)

    def bm25_scores(self, term_ids, want_scores=True, k=0):
        s = np.zeros(self.n_rows)
        for t in term_ids:
            r = self.rows[self.indptr[t]:self.indptr[t + 1]]
            tf = self.vals[self.indptr[t]:self.indptr[t + 1]]
            s[r] += self.idf[t] * (tf * (self.k1 + 1) / (tf + self.k1 * (1 - self.b + self.b * self.doc_len[r] / self.avgdl)))
        self._last = s
        D, I = self._topk(s, k)
        return (s if want_scores else None), D, I

    def tfidf_scores(self, term_ids, weights, want_scores=True, k=0):
        s = np.zeros(self.n_rows)
        for t, w in zip(term_ids, weights):
            r = self.rows[self.indptr[t]:self.indptr[t + 1]]
            s[r] += w * self.vals[self.indptr[t]:self.indptr[t + 1]]
        self._last = s
        D, I = self._topk(s, k)
        return (s if want_scores else None), D, I

    def ranks(self, rows):
        order = np.lexsort((np.arange(self.n_rows), -self._last))
        rk = np.empty(self.n_rows, np.int64)
        rk[order] = np.arange(self.n_rows)
        return rk[np.asarray(rows, np.int64)]

    def gather(self, rows):
        return self._last[np.asarray(rows, np.int64)]

    def count_ahead(self, keys, ids):
        s, j = self._last, np.arange(self.n_rows)
        return np.array([int(((s > k) | ((s == k) & (j < i))).sum()) for k, i in zip(keys, ids)], dtype=np.int64)


def _rrf(ids, lens, k):
    rankings, o = [], 0
    for L in lens:
        rankings.append([(int(i), 0.0) for i in ids[o:o + L]])   # already ordered; equal scores keep order
        o += L
    fused = ofu.reciprocal_rank_fusion(rankings, k=k)
    return np.array(list(fused.keys()), np.int64), np.array(list(fused.values()), np.float64)


def _mmr(cand, q, lam, top_k, byte_len, max_bytes):
    return ofu.mmr_selection(cand, q, list(map(int, byte_len)), lam, top_k, max_bytes)


@contextlib.contextmanager
def oracle_backends():
    from rag_system_amd import bm25, candidates, faiss_compat
    faiss_compat.set_backend_factory(OracleFlatBackend)
    bm25.set_csc_factory(OracleCsc)
    candidates.set_impl(rrf=_rrf, mmr=_mmr,
                        mmr_stored=lambda b, rows, q, lam, top_k, bl, mb: _mmr(b._rows[np.asarray(rows)], q, lam, top_k, bl, mb))
    try:
        yield
    finally:
        faiss_compat.set_backend_factory(None)
        bm25.set_csc_factory(None)
        candidates.set_impl()
