"""TF-IDF cosine scoring on the MI355X.

Reference call sites: search_engine.py:284-285 and :330-331
    query_vec = tfidf_vectorizer.transform([query])
    tfidf_scores = cosine_similarity(query_vec, tfidf_matrix)[0]
followed (rrf) by a stable descending sort of ALL documents (:332-336).

The vocabulary / idf / matrix are built by scikit-learn's TfidfVectorizer exactly as the
reference does (vector_storage.py:216-217) -- that is host-side index construction and
stays in sklearn so tfidf.pkl remains interchangeable.  What moves to the GPU is the per-query
product: cosine_similarity re-normalises both operands and multiplies q (1 x V) with M^T;
here M is re-normalised ONCE at upload (sklearn.preprocessing.normalize, the same routine
cosine_similarity calls), stored term-major in HBM, and each query runs rag_tfidf_scores
(csrc/sparse.hip), which adds w_t * M[doc, t] term by term in scipy's order.
"""
import contextlib
import math
import threading

import numpy as np

from . import bm25 as _bm25


class TfidfScorer:
    def __init__(self, vectorizer, matrix, shard=None):
        """shard=(first, count) (rag_system_amd.sharded): only the document rows [first, first + count) of
        `matrix` are uploaded, as local rows 0 .. count-1 (row normalisation is per row: the block's rows are
        those of the whole matrix)."""
        from sklearn.preprocessing import normalize
        self.vectorizer = vectorizer
        matrix = matrix.tocsr()
        if shard is not None:
            matrix = matrix[int(shard[0]):int(shard[0]) + int(shard[1])]
        self.n_docs, self.n_terms = matrix.shape
        mn = normalize(matrix, copy=True)   # what cosine_similarity does to Y
        csc = mn.tocsc()
        csc.sort_indices()
        self._csc = _bm25.make_csc(self.n_terms, self.n_docs, csc.indptr.astype(np.int64),
                                   csc.indices.astype(np.int32), csc.data.astype(np.float64))
        # one begun query per handle (rag_tfidf_query_begin): queries of different threads (the reference's UI
        # runs one thread per query, UI_manager.py:883) wait for each other here instead of being refused
        self._lock = threading.Lock()
        self._begun_by = None

    @contextlib.contextmanager
    def _serialised(self):
        if self._begun_by == threading.get_ident():
            yield
            return
        with self._lock:
            yield

    @staticmethod
    def _normalize_row(indices, data):
        """What cosine_similarity does to a one-row X -- sklearn.preprocessing.normalize(X) (l2:
        inplace_csr_row_normalize_l2 sums the squares of the stored values one after another, takes
        the square root and divides each value by it; an all-zero row stays as it is) followed by
        sort_indices() -- restated in a few float operations, because building a scipy matrix and
        calling sklearn per query costs more than the GPU side of the whole query (~0.15 ms against
        ~0.03 ms).  Bit-identical to the sklearn route (tests/test_host_logic.py, test_sparse_gpu.py)."""
        indices = np.asarray(indices, dtype=np.int32)
        data = np.asarray(data, dtype=np.float64)
        ssq = 0.0
        for x in data.tolist():
            ssq += x * x
        if ssq != 0.0:
            data = data / math.sqrt(ssq)
        order = np.argsort(indices, kind="stable")
        return np.ascontiguousarray(indices[order]), np.ascontiguousarray(data[order])

    def _normalized(self, qv):
        qv = qv.tocsr()
        return self._normalize_row(qv.indices, qv.data)

    def _query(self, query):
        return self._normalized(self.vectorizer.transform([query]))

    def scores_from_row(self, indices, data):
        """Scores for an already vectorised query row (the output of vectorizer.transform)."""
        t, w = self._normalize_row(indices, data)
        with self._serialised():
            s, _, _ = self._csc.tfidf_scores(t, w, want_scores=True, k=0)
        return s

    def top_k_from_row(self, indices, data, k):
        """top_k() for an already vectorised query row (k <= MAX_FUSED_K)."""
        t, w = self._normalize_row(indices, data)
        with self._serialised():
            _, D, I = self._csc.tfidf_scores(t, w, want_scores=False, k=min(int(k), self.n_docs))
        return I, D

    def scores(self, query):
        """cosine_similarity(vectorizer.transform([query]), matrix)[0] as float64 (n_docs,)."""
        t, w = self._query(query)
        with self._serialised():
            s, _, _ = self._csc.tfidf_scores(t, w, want_scores=True, k=0)
        return s

    def top_k(self, query, k):
        """First k (doc index, score) of the stable descending sort of all documents."""
        k = min(int(k), self.n_docs)
        t, w = self._query(query)
        if k > _bm25.BM25Okapi.MAX_FUSED_K:
            s = self.scores(query)
            ids = np.array(sorted(range(len(s)), key=lambda i: s[i], reverse=True)[:k], dtype=np.int64)
            return ids, s[ids]
        with self._serialised():
            _, D, I = self._csc.tfidf_scores(t, w, want_scores=False, k=k)
        return I, D

    def ranks(self, doc_indices):
        """0-based rank of each document in the stable descending sort of the scores of the
        LAST scores()/top_k() call (search_engine.py:332-336 sorts all documents; RRF only
        needs the ranks of its candidates)."""
        return self._csc.ranks(doc_indices)

    def gather(self, doc_indices):
        """Scores of the given documents from the LAST scores()/top_k() call."""
        return self._csc.gather(doc_indices)

    def count_ahead(self, keys, ids):
        """Per (score, id) pair: how many documents of the LAST scores()/top_k() call stand ahead of it (ids in
        this scorer's row numbering, possibly outside it: rag_csc_count_ahead)."""
        return self._csc.count_ahead(keys, ids)

    def ranks_from_row(self, indices, data, k, doc_indices):
        """(top_idx, top_scores, ranks): top_k_from_row(k) + ranks(doc_indices) in one engine call."""
        t, w = self._normalize_row(indices, data)
        with self._serialised():
            D, I, ranks, _ = self._csc.tfidf_query(t, w, min(int(k), self.n_docs), doc_indices, want_scores=False)
        return I, D, ranks

    def query(self, query, k, doc_indices):
        """(top_idx, top_scores, ranks, scores) = top_k(query, k) + ranks(doc_indices) +
        gather(doc_indices), in one engine call when the backend has one (rag_tfidf_query)."""
        k = min(int(k), self.n_docs)
        fused = getattr(self._csc, "tfidf_query", None)
        if (fused is None or k > _bm25.BM25Okapi.MAX_FUSED_K or not len(doc_indices)
                or len(doc_indices) > getattr(self._csc, "MAX_QUERY_ROWS", 0)):
            # three engine calls that belong together (ranks / gather read the scores of the top_k call)
            with self._serialised():
                prev, self._begun_by = self._begun_by, threading.get_ident()
                try:
                    I, D = self.top_k(query, k)
                    has = len(doc_indices) > 0
                    return I, D, (self.ranks(doc_indices) if has else []), (self.gather(doc_indices) if has else [])
                finally:
                    self._begun_by = prev
        t, w = self._query(query)
        with self._serialised():
            D, I, ranks, sc = fused(t, w, k, doc_indices)
        return I, D, ranks, sc

    # ---- the same in two halves (rag_tfidf_query_begin / _end): scoring needs only the query ------------

    def _begin_row(self, t, w, k_bound):
        k_bound = min(int(k_bound), self.n_docs)
        begin = getattr(self._csc, "tfidf_query_begin", None)
        if begin is None or k_bound < 1 or k_bound > _bm25.BM25Okapi.MAX_FUSED_K:
            return None
        if self._begun_by == threading.get_ident():
            return None                 # this thread already has a begun query: the caller uses query()
        self._lock.acquire()            # held until query_end / query_abort
        try:
            ok = begin(t, w, k_bound)
        except BaseException:
            self._lock.release()
            raise
        if not ok:
            self._lock.release()
            return None
        self._begun_by = threading.get_ident()
        return k_bound

    def _release_begun(self):
        if self._begun_by is not None:
            self._begun_by = None
            self._lock.release()

    def query_begin(self, query, k_bound):
        """Enqueue the scoring of `query` and the selection of its k_bound best documents; returns a token for
        query_end, or None when the engine cannot split the call (use query())."""
        t, w = self._query(query)
        kb = self._begin_row(t, w, k_bound)
        return None if kb is None else ("tfidf-begun", kb)

    def query_end(self, token, k, doc_indices):
        """(top_idx, top_scores, ranks, scores) as query(query, k, doc_indices); k <= the bound given to
        query_begin, len(doc_indices) <= MAX_QUERY_ROWS."""
        kb = token[1]
        k = min(int(k), self.n_docs, kb)
        try:
            if len(doc_indices) > getattr(self._csc, "MAX_QUERY_ROWS", 0):   # more rows than one rank query takes
                D, I, _, _ = self._csc.tfidf_query_end(k, [])
                return I, D, self.ranks(doc_indices), self.gather(doc_indices)
            D, I, ranks, sc = self._csc.tfidf_query_end(k, doc_indices)
            return I, D, ranks, sc
        finally:
            self._release_begun()

    def set_doc_tables(self, flat_row, text_bytes):
        """Side tables for hybrid_tail's MMR stage (rag_csc_set_doc_tables); False when the backend has none."""
        fn = getattr(self._csc, "set_doc_tables", None)
        if fn is None:
            return False
        with self._serialised():
            fn(flat_row, text_bytes)
        return True

    def hybrid_tail(self, token, flat_index, dense_docs, bm25_docs, top_k, k_const=60, threshold=None, use_knee=False,
                    q=None, lam=0.5, max_bytes=None):
        """The rest of a strategy=rrf query behind a begun query in one engine call (rag_hybrid_rrf_mmr): RRF with
        the full TF-IDF ranking, sort, cut to top_k, knee / threshold, MMR on stored rows.
        -> (fused_docs, fused_scores, n_kept, mmr_order or None); None = not served, the query is STILL begun
        (use query_end).  threshold None = no threshold filter."""
        fn = getattr(self._csc, "hybrid_tail", None)
        if fn is None or token is None:
            return None
        backend = getattr(flat_index, "_b", flat_index) if flat_index is not None else None
        if backend is not None and not hasattr(backend, "_h"):
            backend = None
        try:
            out = fn(backend, dense_docs, bm25_docs, k_const, top_k, -float("inf") if threshold is None else threshold,
                     use_knee, q, lam, max_bytes)
        except BaseException:
            self._abort_quietly()
            raise
        if out is not None:
            self._release_begun()
        return out

    def _abort_quietly(self):
        try:
            self._csc.tfidf_query_abort()
        except Exception:
            pass
        finally:
            self._release_begun()

    def query_abort(self, token):
        """Give up a begun query (the handle accepts other calls again)."""
        try:
            self._csc.tfidf_query_abort()
        finally:
            self._release_begun()

    def ranks_from_row_begin(self, indices, data, k_bound):
        t, w = self._normalize_row(indices, data)
        kb = self._begin_row(t, w, k_bound)
        return None if kb is None else ("tfidf-begun", kb)

    def ranks_from_row_end(self, token, k, doc_indices):
        try:
            D, I, ranks, _ = self._csc.tfidf_query_end(min(int(k), self.n_docs, token[1]), doc_indices,
                                                       want_scores=False)
            return I, D, ranks
        finally:
            self._release_begun()
