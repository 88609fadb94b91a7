"""BASELINE config 5 (strategy=rrf on synthetic documents) as a reusable workload: corpus and index
construction, the per-query pipeline through the engine, and the full-size parity check against
oracle/.  Used by benchmarks/bench_hybrid.py, bench.py's `c5` leg and tests/test_hybrid_full_gpu.py.

Synthetic corpus: seed 777, vocabulary V, Zipf(1.1) term draws, document length
lognormal with mean 120 tokens, one chunk per document; queries of 3-8 terms from the same Zipf;
dense rows from the engine's generator (seed 1234), queries seed 4321.
Per query (search_engine.py:320-350 with top_k = 10):
    dense   index.search(q, 9*top_k)  -> first 3*top_k distinct documents     (rag_flat_search)
    bm25    top 3*top_k chunks                                                 (rag_bm25_scores)
    tfidf   scores of all documents; top (top_k + |A|) + ranks of A            (rag_tfidf_query_begin / _end)
    rrf     1/(60+rank) over the three rankings                                (rag_rrf)
    mmr     top_k of the fused top 3*top_k, stored vectors as embeddings       (rag_mmr)
Round 3: everything behind the dense and the BM25 ranking -- the TF-IDF rank query, the fusion, the cut and MMR --
is ONE engine call (rag_hybrid_rrf_mmr, `tail=True`, the default); `tail=False` keeps the three synchronous calls
(rag_rrf, rag_tfidf_query_end, rag_flat_mmr), and tests compare the two routes entry for entry.

oracle/ is imported only inside check(): as the CHECKER of recorded results, never in anything timed.
"""
import time

import numpy as np


def make_corpus(n_docs, vocab, seed):
    rng = np.random.default_rng(seed)
    sigma = 0.4
    lens = np.clip(rng.lognormal(np.log(120.0) - 0.5 * sigma * sigma, sigma, n_docs), 8, 1000).astype(np.int64)
    indptr = np.concatenate([np.zeros(1, np.int64), np.cumsum(lens)])
    w = np.arange(1, vocab + 1, dtype=np.float64) ** -1.1
    cdf = np.cumsum(w / w.sum())
    terms = np.empty(int(indptr[-1]), dtype=np.int32)
    step = 16_000_000
    for lo in range(0, len(terms), step):           # chunked: bounded temporaries
        hi = min(len(terms), lo + step)
        terms[lo:hi] = np.searchsorted(cdf, rng.random(hi - lo)).astype(np.int32)
    np.minimum(terms, vocab - 1, out=terms)
    nq_terms = rng.integers(3, 9, size=4096)
    return indptr, terms, cdf, nq_terms, rng


def tfidf_matrix(n_docs, n_terms, indptr, rows, tf):
    """TfidfVectorizer defaults on integer tokens: idf = ln((1+n)/(1+df)) + 1, l2 rows."""
    from scipy.sparse import csc_matrix
    from sklearn.preprocessing import normalize
    df = np.diff(indptr)
    idf = np.log((1.0 + n_docs) / (1.0 + df)) + 1.0
    m = csc_matrix((tf * np.repeat(idf, df), rows, indptr), shape=(n_docs, n_terms)).tocsr()
    return normalize(m, copy=False), idf


STAGES = ["dense", "bm25", "tfidf", "rrf", "mmr", "tail", "total"]


class HybridWorkload:
    def __init__(self, n_docs=1_000_000, vocab=200_000, dim=384, n_queries=50, top_k=10, log=None):
        from rag_system_amd import faiss_compat as faiss
        from rag_system_amd.bm25 import BM25Okapi
        from rag_system_amd.tfidf import TfidfScorer
        log = log or (lambda m: None)
        self.n, self.V, self.d, self.top_k = n_docs, vocab, dim, top_k
        log("generating %d documents over a %d-term vocabulary" % (n_docs, vocab))
        self.indptr, self.terms, cdf, nq_terms, rng = make_corpus(n_docs, vocab, 777)
        log("%d tokens; building BM25 postings (k1=1.0, b=0.65: config.py:37-38)" % len(self.terms))
        self.bm = BM25Okapi.from_term_ids(self.indptr, self.terms, k1=1.0, b=0.65, keep_postings=True)
        self.p_indptr, self.p_rows, self.p_tf = self.bm.postings
        self.n_terms = len(self.p_indptr) - 1
        self.nnz = len(self.p_rows)
        log("%d distinct terms, %d postings; building TF-IDF matrix" % (self.n_terms, self.nnz))
        self.M, self.tfidf_idf = tfidf_matrix(n_docs, self.n_terms, self.p_indptr, self.p_rows, self.p_tf)
        self.ts = TfidfScorer(None, self.M)
        log("dense index %d x %d" % (n_docs, dim))
        self.index = faiss.IndexFlatIP(dim)
        self.index._b.add_synthetic(n_docs, 1234, 0)
        qgen = faiss.IndexFlatIP(dim)
        qgen._b.add_synthetic(n_queries, 4321, 0)
        self.Qd = qgen.reconstruct_n(0, n_queries)
        del qgen
        self.queries = []                            # raw integer tokens, as a tokenizer would hand them over
        for qi in range(n_queries):
            raw = np.minimum(np.searchsorted(cdf, rng.random(int(nq_terms[qi]))), vocab - 1)
            self.queries.append([int(t) for t in raw])
        self.vocab_map = self.bm._vocab              # raw integer token -> term id
        self.df_all = np.diff(self.p_indptr)
        self._Mc = None
        # one chunk per document: document r's embedding is row r of the dense index; MMR budgets 120 bytes each
        self.tail_ok = self.ts.set_doc_tables(np.arange(n_docs, dtype=np.int64), np.full(n_docs, 120, dtype=np.int64))

    def embeddings(self, ids):
        """Stored dense rows of the given documents (MMR candidate embeddings)."""
        ids = np.asarray(ids, dtype=np.int64)
        out = np.empty((len(ids), self.d), dtype=np.float32)
        for j, i in enumerate(ids):                  # top_k rows: a handful of small copies
            out[j] = self.index.reconstruct_n(int(i), 1)[0]
        return out

    def query(self, qi, tsum=None, sparse_bytes=None, keep_emb=True, tail=True):
        """One strategy=rrf query through the engine; per-stage wall clock accumulated into tsum."""
        from rag_system_amd.candidates import mmr_select_stored, reciprocal_rank_fusion
        top_k, n = self.top_k, self.n
        raw = self.queries[qi]
        tids = np.asarray([self.vocab_map[t] for t in raw if t in self.vocab_map], dtype=np.int64)
        t0 = time.perf_counter()
        # search_faiss(document mode, 3*top_k) begun, search_bm25(3*top_k) beside it, as search_engine.search_rrf does
        pending = self.index.search_begin(self.Qd[qi:qi + 1], 9 * top_k)
        t1a = time.perf_counter()
        # TF-IDF scoring begun too (search_engine._tfidf_begin): it needs the query only; its rank query follows the fusion
        uq, cnt = np.unique(tids, return_counts=True)
        tf_tok = self.ts.ranks_from_row_begin(uq, cnt * self.tfidf_idf[uq], 7 * top_k)
        t1 = time.perf_counter()
        bI, bD = self.bm.top_k(raw, 3 * top_k)
        bm_rank = [int(i) for i in bI]
        t2a = time.perf_counter()
        D, I = self.index.search_end(pending)
        dense_rank = list(dict.fromkeys(int(i) for i in I[0]))[:3 * top_k]
        t2 = time.perf_counter()
        if tail and self.tail_ok and tf_tok is not None:
            # search_engine.py:340-347 + :131-133 in one call: rank query, fusion, sort, cut, MMR on the stored rows
            out = self.ts.hybrid_tail(tf_tok, self.index, dense_rank, bm_rank, top_k, 60, None, False, self.Qd[qi], 0.5, None)
            if out is not None and out[3] is not None:
                fd, fs, _n_kept, sel = out
                t6 = time.perf_counter()
                ids = [int(i) for i in fd]
                order = list(zip(ids, (float(x) for x in fs)))
                emb = self.embeddings(ids) if keep_emb else None
                if tsum is not None:
                    tsum["dense"] += (t1a - t0) + (t2 - t2a)
                    tsum["bm25"] += t2a - t1
                    tsum["tfidf"] += t1 - t1a
                    tsum["tail"] += t6 - t2
                    tsum["total"] += t6 - t0
                if sparse_bytes is not None:
                    sparse_bytes["bm25"] += 12.0 * float(self.df_all[tids].sum()) + 20.0 * n
                    sparse_bytes["tfidf"] += 12.0 * float(self.df_all[uq].sum()) + 16.0 * n
                return dict(dense=dense_rank, bm25=(bm_rank, bD), tfidf=None, cand=None, ranks=None, fused=order,
                            mmr=[ids[i] for i in sel], tids=tids, emb=emb)
            if out is not None:     # fusion done, MMR not (cannot happen with the tables above)
                raise RuntimeError("rag_hybrid_rrf_mmr left the MMR stage to the caller")
        fused = reciprocal_rank_fusion([dense_rank, bm_rank], k=60)
        t3 = time.perf_counter()
        # TF-IDF: every document is ranked (search_engine.py:332-336); only the candidates and
        # the head of the ranking can reach the fused top_k (rag_system_amd/search_engine.py)
        cand = list(fused)
        if tf_tok is not None:
            tI, tD, ranks = self.ts.ranks_from_row_end(tf_tok, top_k + len(cand), np.asarray(cand, dtype=np.int64))
        else:
            tI, tD, ranks = self.ts.ranks_from_row(uq, cnt * self.tfidf_idf[uq], top_k + len(cand),
                                                   np.asarray(cand, dtype=np.int64))
        t4 = time.perf_counter()
        for doc, rk in zip(cand, ranks):
            fused[doc] = fused[doc] + 1.0 / (60 + int(rk))
        for rank, doc in enumerate(int(i) for i in tI):
            if doc not in fused:
                fused[doc] = 0 + 1.0 / (60 + rank)
        order = sorted(fused.items(), key=lambda kv: kv[1], reverse=True)[:top_k]
        t5 = time.perf_counter()
        ids = [doc for doc, _ in order]
        t5b = t5
        # search_engine.py:131-133 with the stored rows as embeddings (rag_flat_mmr: gathered in HBM)
        sel = mmr_select_stored(self.index, ids, self.Qd[qi], [120] * len(ids), 0.5, top_k, None)
        t6 = time.perf_counter()
        emb = self.embeddings(ids) if keep_emb else None
        if tsum is not None:
            tsum["dense"] += (t1a - t0) + (t2 - t2a)     # enqueue + what was left to wait for
            tsum["bm25"] += t2a - t1                      # (runs beside the dense pass)
            tsum["tfidf"] += (t1 - t1a) + (t4 - t3)       # enqueue of the scoring + the rank query behind the fusion
            tsum["rrf"] += (t3 - t2) + (t5 - t4)
            tsum["mmr"] += t6 - t5b
            tsum["total"] += (t6 - t0) - (t5b - t5)
        if sparse_bytes is not None:
            # fused path (csrc/sparse.hip): a posting is read once (row 4 + val 8 bytes) and its score lives in
            # LDS; per row: doc_len 4 (BM25), the score written once (8) and read once by the selection (8)
            sparse_bytes["bm25"] += 12.0 * float(self.df_all[tids].sum()) + 20.0 * n
            sparse_bytes["tfidf"] += 12.0 * float(self.df_all[uq].sum()) + 16.0 * n
        return dict(dense=dense_rank, bm25=(bm_rank, bD), tfidf=([int(i) for i in tI], tD), cand=cand,
                    ranks=[int(r) for r in ranks], fused=order, mmr=[ids[i] for i in sel], tids=tids, emb=emb)

    def check(self, qis, results):
        """Full-size parity of recorded results against oracle/ (CHECKER role only): BM25 ids and
        float64 scores bit-exact, TF-IDF ids exact and scores to 1e-12, ranks exact, dense ids exact,
        fused top_k identical to RRF over the three FULL rankings (search_engine.py:352-359), MMR
        order identical."""
        from oracle import dense as od
        from oracle import fusion as ofu
        from oracle import sparse as osp
        n, d, top_k = self.n, self.d, self.top_k
        if self._Mc is None:
            self._Mc = self.M.tocsc()
            self._Mc.sort_indices()
        Mc = self._Mc
        all_ids = np.arange(n)
        Qs = np.ascontiguousarray(self.Qd[list(qis)])
        _Dr, Ir, _D64 = od.flat_search_synth(1234, 0, n, d, Qs, 9 * top_k)
        for j, (qi, r) in enumerate(zip(qis, results)):
            tids = r["tids"]
            s_ref = osp.bm25_scores_postings(self.p_indptr, self.p_rows, self.p_tf, self.bm._idf_arr,
                                             np.asarray(self.bm.doc_len), self.bm.avgdl, 1.0, 0.65, tids)
            top = np.lexsort((all_ids, -s_ref))[:3 * top_k]
            assert top.tolist() == r["bm25"][0], "BM25 ids differ from the restatement (query %d)" % qi
            assert s_ref[top].tobytes() == np.asarray(r["bm25"][1]).tobytes(), "BM25 scores differ bitwise"
            uq, cnt = np.unique(tids, return_counts=True)
            w = cnt * self.tfidf_idf[uq]
            w = w / np.sqrt((w * w).sum())
            t_ref = osp.tfidf_scores_postings(Mc.indptr, Mc.indices, Mc.data, uq, w, n)
            t_order = np.lexsort((all_ids, -t_ref))
            if r["tfidf"] is not None:      # (the one-call tail keeps these stages on the device; their effect is
                #                              checked through the fused list below, which needs every rank right)
                assert t_order[:len(r["tfidf"][0])].tolist() == r["tfidf"][0], "TF-IDF ids differ (query %d)" % qi
                assert np.allclose(t_ref[t_order[:len(r["tfidf"][0])]], r["tfidf"][1], rtol=1e-12, atol=0), \
                    "TF-IDF scores differ"
                full_rank = np.empty(n, dtype=np.int64)
                full_rank[t_order] = all_ids
                assert full_rank[r["cand"]].tolist() == r["ranks"], "TF-IDF ranks differ (query %d)" % qi
            assert list(dict.fromkeys(int(i) for i in Ir[j]))[:3 * top_k] == r["dense"], \
                "dense ids differ (query %d)" % qi
            fused = ofu.reciprocal_rank_fusion([[(i, 0.0) for i in r["dense"]], [(i, 0.0) for i in r["bm25"][0]],
                                                [(int(i), 0.0) for i in t_order]], k=60)
            order = sorted(fused.items(), key=lambda kv: kv[1], reverse=True)[:top_k]
            assert order == r["fused"], "fused top_k differs from RRF over the full rankings (query %d)" % qi
            ids = [a for a, _ in order]
            emb = np.stack([od.synth_rows(1234, int(i), 1, d)[0] for i in ids])
            if r["emb"] is not None:
                assert emb.tobytes() == r["emb"].tobytes(), "stored rows differ from the generator (query %d)" % qi
            sel = ofu.mmr_selection(emb, self.Qd[qi], [120] * len(ids), 0.5, top_k, None)
            assert [ids[i] for i in sel] == r["mmr"], "MMR order differs (query %d)" % qi
        return len(qis)


# --------------------------------------------------------------------------------------------- product class

def token_texts(indptr, terms):
    """Document texts for integer-token corpora: token 123 -> the word "t000123", words joined by blanks (what a
    chunker would hand to add_vector); vectorised, one ASCII buffer for the whole corpus."""
    n_tok = len(terms)
    buf = np.empty((n_tok, 8), dtype=np.uint8)
    buf[:, 0] = ord("t")
    buf[:, 7] = ord(" ")
    v = terms.astype(np.int32)
    for j in range(6, 0, -1):
        buf[:, j] = (48 + v % 10).astype(np.uint8)
        v = v // 10
    flat = buf.reshape(-1).tobytes()
    return [flat[8 * int(a):8 * int(b) - 1].decode("ascii") for a, b in zip(indptr[:-1], indptr[1:])]


class _TokenVectorizer:
    """Stand-in for the fitted sklearn TfidfVectorizer of such a store: transform([text]) -> the 1 x n_terms CSR
    row of tf * idf weights (columns = term ids).  The row is left un-normalised (cosine_similarity normalises the
    query anyway, search_engine.py:285), which keeps the weights identical to HybridWorkload.query's."""

    def __init__(self, vocab, idf, n_terms):
        self.vocab, self.idf, self.n_terms = vocab, idf, n_terms

    def transform(self, texts):
        from scipy.sparse import csr_matrix
        ids = [self.vocab[t] for t in texts[0].lower().split() if t in self.vocab]
        uq, cnt = np.unique(np.asarray(ids, dtype=np.int64), return_counts=True)
        return csr_matrix((cnt * self.idf[uq], uq.astype(np.int32), np.array([0, len(uq)], dtype=np.int32)),
                          shape=(1, self.n_terms))


class _LookupEmbeddingManager:
    """embed_query by table: the synthetic queries' dense vectors are given (producing embeddings is outside the
    hot path)."""
    model_name, chunk_size, overlap = "synthetic", 100, 20

    def __init__(self, table):
        self._table = table

        class _NoModel:
            def encode(self, texts, *a, **k):
                raise RuntimeError("the synthetic store has no text encoder (MMR runs on the stored rows)")
        self.model = _NoModel()

    def embed_query(self, query):
        return self._table[query]


def make_search_engine(wl, log=None):
    """rag_system_amd.VectorStorage + SearchEngine over the SAME indices as `wl` (one chunk per document, texts of
    integer-token words), so that search_with_strategy("rrf", ...) can be timed and compared with wl.query()."""
    import copy
    from rag_system_amd.search_engine import SearchEngine
    from rag_system_amd.vector_storage import VectorStorage
    log = log or (lambda m: None)
    n = wl.n
    log("product store: %d document texts" % n)
    texts = token_texts(wl.indptr, wl.terms)
    ids = ["doc%07d" % i for i in range(n)]
    store = VectorStorage(wl.d, embedder="synthetic")
    store.index = wl.index
    store.doc_ids = ids
    store.chunks_data = [{"doc_id": d, "chunk": t, "source": None} for d, t in zip(ids, texts)]
    store.documents = {d: {"chunks": [{"chunk": t, "source": None}]} for d, t in zip(ids, texts)}
    bm = copy.copy(wl.bm)                       # the same postings handle, vocabulary keyed by the words of the texts
    bm._vocab = {"t%06d" % w: t for w, t in wl.bm._vocab.items()}
    store.bm25, store.bm25_doc_ids, store.bm25_indexed_count = bm, store.chunks_data, n
    vec = _TokenVectorizer(bm._vocab, wl.tfidf_idf, wl.n_terms)
    store.tfidf_index, store.tfidf_vectorizer, store.tfidf_doc_ids = wl.M, vec, ids
    wl.ts.vectorizer = vec
    store._tfidf_scorer, store._tfidf_scorer_key = wl.ts, (id(vec), id(wl.M))
    qtexts = [" ".join("t%06d" % t for t in raw) for raw in wl.queries]
    em = _LookupEmbeddingManager({t: wl.Qd[i] for i, t in enumerate(qtexts)})
    return SearchEngine(store, em), store, qtexts


def time_search_engine(wl, eng, qtexts, n_queries, top_k, expect=None):
    """search_with_strategy(q, "rrf", top_k, use_mmr=True, mmr_reuse_index_vectors=True) per query: wall clock,
    then a second pass with per-stage timers around the engine's own methods.  expect[qi] = wl.query(qi): the
    returned documents and scores must be the workload's (same indices, same kernels)."""
    kw = dict(strategy="rrf", top_k=top_k, use_mmr=True, mmr_lambda=0.5, mmr_reuse_index_vectors=True)
    for qi in range(min(5, n_queries)):
        eng.search_with_strategy(qtexts[qi], **kw)
    t0 = time.perf_counter()
    res = [eng.search_with_strategy(qtexts[qi], **kw) for qi in range(n_queries)]
    total = time.perf_counter() - t0
    if expect is not None:
        for qi, r in expect.items():
            got = [int(x["doc_id"][3:]) for x in res[qi]]
            assert got == r["mmr"], "SearchEngine returns other documents than the workload (query %d)" % qi
            sc = dict(r["fused"])
            assert all(x["score"] == sc[int(x["doc_id"][3:])] for x in res[qi]), "fused scores differ (query %d)" % qi
    # per-stage pass (timers add a little; reported beside, not inside, the wall clock above)
    acc = {}

    def timed(obj, name, key):
        fn = getattr(obj, name)

        def wrapper(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        setattr(obj, name, wrapper)
        return fn
    saved = [(eng, "_search_faiss_begin", timed(eng, "_search_faiss_begin", "embed + dense begin")),
             (eng, "_tfidf_begin", timed(eng, "_tfidf_begin", "tfidf vectorise + begin")),
             (eng, "search_bm25", timed(eng, "search_bm25", "bm25 (tokenise, call, hit dicts)")),
             (eng, "_search_faiss_end", timed(eng, "_search_faiss_end", "dense end (wait, hit dicts, group by doc)")),
             (eng, "_rrf_tail", timed(eng, "_rrf_tail", "tail (rows, one call, result dicts)"))]
    t1 = time.perf_counter()
    for qi in range(n_queries):
        eng.search_with_strategy(qtexts[qi], **kw)
    total2 = time.perf_counter() - t1
    for obj, name, fn in saved:
        setattr(obj, name, fn)
    stage_ms = {k: v / n_queries * 1e3 for k, v in acc.items()}
    stage_ms["other (dispatch, sort, knee / threshold)"] = (total2 - sum(acc.values())) / n_queries * 1e3
    return {"value": n_queries / total, "unit": "QPS", "ms_per_query": total / n_queries * 1e3, "stage_ms": stage_ms,
            "call": "SearchEngine.search_with_strategy(q, 'rrf', %d, use_mmr=True, mmr_reuse_index_vectors=True) on a "
                    "VectorStorage over the same indices (1 chunk per document, texts of integer-token words; "
                    "query vectors by lookup, a light stand-in for the fitted TfidfVectorizer)" % top_k}


# --------------------------------------------------------------------------------------------- CPU baseline

def cpu_baseline(wl, sample_docs=50_000, n_queries=5, log=None):
    """The reference's own per-query work (rank_bm25 dict loop + Python sorts over all documents + sklearn cosine
    + faiss-style flat scan) restated on the first `sample_docs` documents, one core (oracle/: baseline role)."""
    from oracle import dense as od
    from oracle import sparse as osp
    from scipy.sparse import csr_matrix
    from sklearn.feature_extraction.text import TfidfTransformer
    from sklearn.metrics.pairwise import cosine_similarity
    (log or (lambda m: None))("CPU leg: restatement on the first %d documents" % sample_docs)
    indptr, terms, top_k = wl.indptr, wl.terms, wl.top_k
    m = min(sample_docs, wl.n)
    corpus = [terms[indptr[i]:indptr[i + 1]].tolist() for i in range(m)]
    ref = osp.BM25OkapiRef(corpus, k1=1.0, b=0.65)
    X = od.synth_rows(1234, 0, m, wl.d)
    lens = np.diff(indptr[:m + 1])
    counts = csr_matrix((np.ones(int(lens.sum())), terms[:int(indptr[m])], indptr[:m + 1]), shape=(m, wl.V))
    counts.sum_duplicates()
    tr = TfidfTransformer()
    Mtf = tr.fit_transform(counts)
    nq = min(n_queries, len(wl.queries))
    t0 = time.perf_counter()
    for qi in range(nq):
        raw = wl.queries[qi]
        s = ref.get_scores(raw)
        sorted(range(len(s)), key=lambda i: s[i], reverse=True)[:3 * top_k]
        qv = tr.transform(csr_matrix((np.ones(len(raw)), (np.zeros(len(raw), int), raw)), shape=(1, wl.V)))
        sc = cosine_similarity(qv, Mtf)[0]
        sorted(((i, float(sc[i])) for i in range(m)), key=lambda x: x[1], reverse=True)
        od.cpu_baseline_search(X, wl.Qd[qi:qi + 1], 9 * top_k, 1)
    el = time.perf_counter() - t0
    return {"value": nq / el * (m / float(wl.n)), "unit": "QPS", "cores": 1, "kind": "port",
            "sample": "%d queries over the first %d of %d documents (%.1f s): rank_bm25-style dict loop + Python "
                      "sorts over all documents + sklearn cosine + flat fp32 scan; QPS scaled by the document "
                      "ratio (every term is linear in the document count)" % (nq, m, wl.n, el)}
