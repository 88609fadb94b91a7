"""BM25Okapi with rank_bm25's API, scored on the MI355X.

Reference call sites: BM25Okapi(corpus, k1=, b=) vector_storage.py:170, BM25Okapi(corpus)
:305 (library defaults k1=1.5, b=0.75, epsilon=0.25), bm25.get_scores(tokens) :178 followed by
the full stable descending sort of :180.

The index statistics equal rank_bm25's: tokens are mapped to integer words on the host, the
vocabulary order, term frequencies and document frequencies come from rag_postings_build
(csrc/build.hip: sort + segment on the GPU), idf with the epsilon floor and avgdl are computed on
the host in rank_bm25's order; the postings are uploaded once as a term-major (CSC) matrix;
get_scores / top_k run rag_bm25_scores (csrc/sparse.hip).
"""
import ctypes as C
import math

import numpy as np

from . import _lib


class _HipCsc:
    """rag_csc_t handle (include/rag_mi355x.h)."""

    def __init__(self, n_terms, n_rows, indptr, rows, vals):
        self._lib = _lib.engine()
        self.n_terms, self.n_rows = int(n_terms), int(n_rows)
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        h = C.c_void_p()
        _lib.check(self._lib.rag_csc_create(self.n_terms, self.n_rows, _lib.ptr(indptr), _lib.ptr(rows),
                                            _lib.ptr(vals), C.byref(h)), self._lib)
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.rag_csc_free(h)
            except Exception:
                pass

    def bm25_set(self, idf, doc_len, avgdl, k1, b):
        idf = np.ascontiguousarray(idf, dtype=np.float64)
        doc_len = np.ascontiguousarray(doc_len, dtype=np.int32)
        _lib.check(self._lib.rag_bm25_set(self._h, _lib.ptr(idf), _lib.ptr(doc_len), float(avgdl), float(k1),
                                          float(b)), self._lib)

    def _out(self, want_scores, k):
        scores = np.empty(self.n_rows, dtype=np.float64) if want_scores else None
        D = np.empty(max(k, 1), dtype=np.float64)
        I = np.empty(max(k, 1), dtype=np.int64)
        return scores, D, I

    def bm25_scores(self, term_ids, want_scores=True, k=0):
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        scores, D, I = self._out(want_scores, k)
        _lib.check(self._lib.rag_bm25_scores(self._h, len(t), _lib.ptr(t), _lib.ptr(scores), int(k), _lib.ptr(D),
                                             _lib.ptr(I)), self._lib)
        return scores, D[:k], I[:k]

    def tfidf_scores(self, term_ids, weights, want_scores=True, k=0):
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        scores, D, I = self._out(want_scores, k)
        _lib.check(self._lib.rag_tfidf_scores(self._h, len(t), _lib.ptr(t), _lib.ptr(w), _lib.ptr(scores), int(k),
                                              _lib.ptr(D), _lib.ptr(I)), self._lib)
        return scores, D[:k], I[:k]

    def ranks(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty(len(r), dtype=np.int64)
        _lib.check(self._lib.rag_csc_ranks(self._h, len(r), _lib.ptr(r), _lib.ptr(out)), self._lib)
        return out


    def gather(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty(len(r), dtype=np.float64)
        _lib.check(self._lib.rag_csc_gather(self._h, len(r), _lib.ptr(r), _lib.ptr(out)), self._lib)
        return out

    def count_ahead(self, keys, ids):
        """rag_csc_count_ahead: per (score, id) pair, how many rows of the last score vector stand ahead of it
        (ids in this index' row numbering, possibly outside it: another shard's rows)."""
        kk = np.ascontiguousarray(keys, dtype=np.float64)
        ii = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty(len(kk), dtype=np.int64)
        _lib.check(self._lib.rag_csc_count_ahead(self._h, len(kk), _lib.ptr(kk), _lib.ptr(ii), _lib.ptr(out)), self._lib)
        return out

    MAX_QUERY_ROWS = 1024

    def tfidf_query(self, term_ids, weights, k, rows, want_scores=True):
        """(D, I, ranks, row_scores): tfidf_scores(k) + ranks(rows) + gather(rows) in one call
        (rag_tfidf_query; len(rows) <= MAX_QUERY_ROWS).  row_scores is None unless wanted."""
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        r = np.ascontiguousarray(rows, dtype=np.int64)
        _, D, I = self._out(False, k)
        ranks = np.empty(len(r), dtype=np.int64)
        sc = np.empty(len(r), dtype=np.float64) if want_scores else None
        _lib.check(self._lib.rag_tfidf_query(self._h, len(t), _lib.ptr(t), _lib.ptr(w), int(k), _lib.ptr(D), _lib.ptr(I),
                                             len(r), _lib.ptr(r), _lib.ptr(ranks), _lib.ptr(sc)), self._lib)
        return D[:k], I[:k], ranks, sc


    def tfidf_query_begin(self, term_ids, weights, k):
        """First half of tfidf_query (rag_tfidf_query_begin): scoring and the top-k selection enqueued on the
        handle's own stream.  Returns False when the engine's one-launch path does not serve this index / k
        (nothing is begun then: call tfidf_query)."""
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        rc = self._lib.rag_tfidf_query_begin(self._h, len(t), _lib.ptr(t), _lib.ptr(w), int(k))
        if rc == _lib.RAG_ERR_UNSUPPORTED:
            return False
        _lib.check(rc, self._lib)
        return True

    def tfidf_query_abort(self):
        """Ends a begun query without reading anything (error paths)."""
        self._lib.rag_tfidf_query_end(self._h, 0, None, None, 0, None, None, None)

    def tfidf_query_end(self, k, rows, want_scores=True):
        """Second half: (D, I, ranks, row_scores) as tfidf_query, k <= the k the query was begun with."""
        r = np.ascontiguousarray(rows, dtype=np.int64)
        _, D, I = self._out(False, k)
        ranks = np.empty(len(r), dtype=np.int64)
        sc = np.empty(len(r), dtype=np.float64) if want_scores else None
        _lib.check(self._lib.rag_tfidf_query_end(self._h, int(k), _lib.ptr(D), _lib.ptr(I), len(r), _lib.ptr(r),
                                                 _lib.ptr(ranks), _lib.ptr(sc)), self._lib)
        return D[:k], I[:k], ranks, sc


    def set_doc_tables(self, flat_row, text_bytes):
        """rag_csc_set_doc_tables: per row, the flat-index row holding the document's embedding (-1: none) and
        the UTF-8 length of its text; None, None drops the tables."""
        if flat_row is None or text_bytes is None:
            _lib.check(self._lib.rag_csc_set_doc_tables(self._h, None, None), self._lib)
            return
        fr = np.ascontiguousarray(flat_row, dtype=np.int64)
        tb = np.ascontiguousarray(text_bytes, dtype=np.int64)
        if len(fr) != self.n_rows or len(tb) != self.n_rows:
            raise ValueError("doc tables need one entry per row (%d)" % self.n_rows)
        _lib.check(self._lib.rag_csc_set_doc_tables(self._h, _lib.ptr(fr), _lib.ptr(tb)), self._lib)

    def hybrid_tail(self, flat_backend, dense_docs, bm25_docs, k_const, top_k, threshold, use_knee, q, lam, max_bytes):
        """rag_hybrid_rrf_mmr on this handle's begun query.  -> (fused_docs, fused_scores, n_kept, order or None),
        or None when the engine does not serve the shape (the query stays begun: tfidf_query_end).
        (Output buffers and their pointers are kept per handle: the caller holds the begun query, so calls on
        one handle never overlap; a ctypes call with six fresh numpy pointers costs more than the kernel's launch.)"""
        buf = getattr(self, "_tail_buf", None)
        if buf is None or buf[0] < top_k:
            cap = max(64, int(top_k))
            fd, fs, order = np.empty(cap, np.int64), np.empty(cap, np.float64), np.empty(cap, np.int32)
            ints = (C.c_int * 4)()
            buf = self._tail_buf = (cap, fd, fs, order, ints, _lib.ptr(fd), _lib.ptr(fs), _lib.ptr(order),
                                    C.cast(ints, C.c_void_p))
        _cap, fd, fs, order, ints, pfd, pfs, porder, pints = buf
        nd, nb = len(dense_docs), len(bm25_docs)
        dd = (C.c_int64 * max(nd, 1))(*dense_docs)
        bd = (C.c_int64 * max(nb, 1))(*bm25_docs)
        qv = None if q is None else np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        base = pints.value
        rc = self._lib.rag_hybrid_rrf_mmr(
            self._h, flat_backend._h if flat_backend is not None else None, nd, dd, nb, bd,
            float(k_const), int(top_k), float(threshold), 1 if use_knee else 0, _lib.ptr(qv), float(lam),
            -1 if max_bytes is None else int(max_bytes), base, pfd, pfs, base + 4, base + 8, base + 12, porder)
        if rc == _lib.RAG_ERR_UNSUPPORTED:
            return None
        _lib.check(rc, self._lib)
        nf, nk, done, ns = ints[0], ints[1], ints[2], ints[3]
        return (fd[:nf].copy(), fs[:nf].copy(), nk, order[:ns].tolist() if done else None)


_csc_factory = _HipCsc


def set_csc_factory(factory):
    """Test seam (host-logic tests inject the CPU oracle); None restores the HIP engine."""
    global _csc_factory
    _csc_factory = factory or _HipCsc


def make_csc(n_terms, n_rows, indptr, rows, vals):
    return _csc_factory(n_terms, n_rows, indptr, rows, vals)


def device_postings(doc_indptr, words, n_words):
    """(term_word, indptr, rows, tf) of a corpus given as integer words (document i =
    words[doc_indptr[i]:doc_indptr[i+1]], words in [0, n_words)), built on the GPU by
    rag_postings_build (csrc/build.hip: sort + segment).  Terms are numbered by first appearance;
    term_word[t] is the word of term t; the postings of a term ascend by document."""
    lib = _lib.engine()
    doc_indptr = np.ascontiguousarray(doc_indptr, dtype=np.int64)
    words = np.ascontiguousarray(words, dtype=np.int32)
    n_docs, n_tokens, n_words = len(doc_indptr) - 1, len(words), int(n_words)
    n_terms, nnz = C.c_int64(0), C.c_int64(0)
    term_word = np.empty(max(n_words, 1), dtype=np.int32)
    indptr = np.empty(n_words + 1, dtype=np.int64)
    rows = np.empty(max(n_tokens, 1), dtype=np.int32)
    tf = np.empty(max(n_tokens, 1), dtype=np.float64)
    _lib.check(lib.rag_postings_build(n_docs, _lib.ptr(doc_indptr), n_tokens, _lib.ptr(words), n_words,
                                      C.byref(n_terms), _lib.ptr(term_word), _lib.ptr(indptr), _lib.ptr(rows),
                                      _lib.ptr(tf), C.byref(nnz)), lib)
    nt, nz = n_terms.value, nnz.value
    return term_word[:nt], indptr[:nt + 1], rows[:nz].copy(), tf[:nz].copy()


class BM25Okapi:
    """rank_bm25.BM25Okapi(corpus, tokenizer=None, k1=1.5, b=0.75, epsilon=0.25)."""

    MAX_FUSED_K = _lib.MAX_K

    def __init__(self, corpus, tokenizer=None, k1=1.5, b=0.75, epsilon=0.25, shard=None):
        """shard=(first, count) (not in rank_bm25; rag_system_amd.sharded): the statistics -- idf, avgdl -- are
        those of the WHOLE corpus, but only the postings of documents [first, first + count) are uploaded, as
        local rows 0 .. count-1 (get_scores / top_k then cover this block only)."""
        self.k1 = k1
        self.b = b
        self.epsilon = epsilon
        self.tokenizer = tokenizer
        self._shard = shard
        if tokenizer:
            corpus = [tokenizer(doc) for doc in corpus]
        self.corpus_size = 0
        self.avgdl = 0
        self.doc_len = []
        self.idf = {}
        self._vocab = {}          # word -> term id, in first-appearance order (= nd order)
        if _csc_factory is _HipCsc:
            # device build: tokens -> integer words in one pass, postings by sort + segment
            # on the GPU instead of a dictionary of frequencies per document
            vocab, words, indptr = self._vocab, [], [0]
            for document in corpus:
                for word in document:
                    t = vocab.get(word)
                    if t is None:
                        t = len(vocab)
                        vocab[word] = t
                    words.append(t)
                indptr.append(len(words))
            self.corpus_size = len(indptr) - 1
            if self.corpus_size == 0:
                raise ZeroDivisionError("division by zero")  # rank_bm25: num_doc / corpus_size
            lens = np.diff(np.asarray(indptr, dtype=np.int64))
            self.doc_len = lens.tolist()
            self.avgdl = len(words) / self.corpus_size
            term_word, p_indptr, rows, tf = device_postings(indptr, np.asarray(words, dtype=np.int32), len(vocab))
            assert np.array_equal(term_word, np.arange(len(vocab)))   # words were numbered by first appearance
            self._finish(np.diff(p_indptr), p_indptr, rows, tf)
            return
        post_rows = []            # per term: list of doc ids (ascending)
        post_tf = []
        num_doc = 0
        for di, document in enumerate(corpus):
            self.doc_len.append(len(document))
            num_doc += len(document)
            frequencies = {}
            for word in document:
                frequencies[word] = frequencies.get(word, 0) + 1
            for word, freq in frequencies.items():
                t = self._vocab.get(word)
                if t is None:
                    t = len(self._vocab)
                    self._vocab[word] = t
                    post_rows.append([])
                    post_tf.append([])
                post_rows[t].append(di)
                post_tf[t].append(freq)
            self.corpus_size += 1
        if self.corpus_size == 0:
            raise ZeroDivisionError("division by zero")  # rank_bm25: num_doc / corpus_size
        self.avgdl = num_doc / self.corpus_size
        df = np.fromiter((len(r) for r in post_rows), dtype=np.int64, count=len(post_rows))
        self._finish(df,
                     np.concatenate([np.zeros(1, np.int64), np.cumsum(df)]),
                     np.fromiter((d for r in post_rows for d in r), dtype=np.int32, count=int(df.sum())),
                     np.fromiter((f for r in post_tf for f in r), dtype=np.float64, count=int(df.sum())))

    @classmethod
    def from_term_ids(cls, doc_indptr, doc_terms, k1=1.5, b=0.75, epsilon=0.25, keep_postings=False):
        """Corpus given as integer tokens: document i = doc_terms[doc_indptr[i]:doc_indptr[i+1]].
        Same statistics as the list-of-tokens constructor (token = the integer), built with
        numpy so that million-document synthetic corpora index in seconds.
        keep_postings: keep the host copy of the term-major postings as self.postings =
        (indptr, rows, tf) (benchmarks cross-check full-size results with it)."""
        self = cls.__new__(cls)
        self._shard = None
        self._keep_postings = keep_postings
        self.k1, self.b, self.epsilon, self.tokenizer = k1, b, epsilon, None
        doc_indptr = np.asarray(doc_indptr, dtype=np.int64)
        doc_terms = np.asarray(doc_terms, dtype=np.int64)
        n = len(doc_indptr) - 1
        if n <= 0:
            raise ZeroDivisionError("division by zero")
        self.corpus_size = n
        lens = np.diff(doc_indptr)
        self.doc_len = lens.tolist()
        self.avgdl = int(lens.sum()) / n
        if (_csc_factory is _HipCsc and len(doc_terms) > 0 and int(doc_terms.min()) >= 0
                and int(doc_terms.max()) < (1 << 27)):
            # device build: first-appearance vocabulary, (term, doc) pairs and frequencies by sort + segment
            term_word, p_indptr, rows, tf = device_postings(doc_indptr, doc_terms.astype(np.int32), int(doc_terms.max()) + 1)
            self._vocab = {int(w): t for t, w in enumerate(term_word.tolist())}
            self.idf = {}
            self._finish(np.diff(p_indptr), p_indptr, rows, tf)
            return self
        doc_of = np.repeat(np.arange(n, dtype=np.int64), lens)
        # vocabulary in first-appearance order
        uniq, first_pos = np.unique(doc_terms, return_index=True)
        order = np.argsort(first_pos, kind="stable")
        tid_of_uniq = np.empty(len(uniq), dtype=np.int64)
        tid_of_uniq[order] = np.arange(len(uniq))
        tids = tid_of_uniq[np.searchsorted(uniq, doc_terms)]
        self._vocab = {int(w): int(t) for w, t in zip(uniq[order], range(len(uniq)))}
        # (term, doc) pairs -> term frequencies
        key = tids * n + doc_of
        ukey, tf = np.unique(key, return_counts=True)
        p_term = ukey // n
        p_doc = (ukey - p_term * n).astype(np.int32)
        df = np.bincount(p_term, minlength=len(uniq)).astype(np.int64)
        self.idf = {}
        self._finish(df, np.concatenate([np.zeros(1, np.int64), np.cumsum(df)]), p_doc, tf.astype(np.float64))
        return self

    def _finish(self, df, indptr, rows, tf):
        n = self.corpus_size
        words = list(self._vocab.keys())
        idf = np.empty(len(words), dtype=np.float64)
        idf_sum = 0
        negative = []
        for t in range(len(words)):
            freq = int(df[t])
            v = math.log(n - freq + 0.5) - math.log(freq + 0.5)
            idf[t] = v
            idf_sum += v
            if v < 0:
                negative.append(t)
        self.average_idf = idf_sum / len(words)   # ZeroDivisionError on an empty vocabulary, as rank_bm25
        eps = self.epsilon * self.average_idf
        for t in negative:
            idf[t] = eps
        self.idf = {w: float(idf[t]) for t, w in enumerate(words)}
        self._idf_arr = idf
        if getattr(self, "_keep_postings", False):
            self.postings = (indptr, rows, tf)
        doc_len = np.asarray(self.doc_len, dtype=np.int32)
        self.n_local = n
        shard = getattr(self, "_shard", None)
        if shard is not None:       # this rank's block of the rows: filter the (term-major, row-ascending) postings
            lo, cnt = int(shard[0]), int(shard[1])
            rows = np.asarray(rows)
            keep = (rows >= lo) & (rows < lo + cnt)
            csum = np.concatenate([np.zeros(1, np.int64), np.cumsum(keep, dtype=np.int64)])
            indptr = csum[np.asarray(indptr, dtype=np.int64)]
            rows = (rows[keep] - lo).astype(np.int32)
            tf = np.asarray(tf)[keep]
            doc_len = doc_len[lo:lo + cnt]
            self.n_local = cnt
            if cnt == 0:
                self._csc = None
                return
        self._csc = make_csc(len(words), self.n_local, indptr, rows, tf)
        self._csc.bm25_set(idf, doc_len, self.avgdl, self.k1, self.b)

    def _term_ids(self, query):
        # out-of-vocabulary tokens add (idf.get(q) or 0) * 0 = 0: dropped.  Repeats stay.
        v = self._vocab
        return [v[q] for q in query if q in v]

    def get_scores(self, query):
        """np.float64 (corpus_size,) -- rank_bm25.BM25Okapi.get_scores."""
        scores, _, _ = self._csc.bm25_scores(self._term_ids(query), want_scores=True, k=0)
        return scores

    def top_k(self, query, k):
        """(ids, scores) of the first k entries of sorted(range(n), key=scores, reverse=True)
        (vector_storage.py:180) without materialising the score vector on the host."""
        k = min(int(k), getattr(self, "n_local", self.corpus_size))
        if k <= 0:
            return np.empty(0, np.int64), np.empty(0, np.float64)
        if k > self.MAX_FUSED_K:
            scores = self.get_scores(query)
            ids = np.array(sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)[:k], dtype=np.int64)
            return ids, scores[ids]
        _, D, I = self._csc.bm25_scores(self._term_ids(query), want_scores=False, k=k)
        return I, D
