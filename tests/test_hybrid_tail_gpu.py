"""rag_hybrid_rrf_mmr (csrc/cand.hip hybrid_tail_kernel): the tail of a strategy=rrf query in one call --
reciprocal_rank_fusion over the three rankings + sort + cut (search_engine.py:340-347, :352-359), knee / threshold
(:121-129), MMR (:396-468).  Checked bit for bit against (a) the three separate engine calls it replaces
(rag_rrf + rag_tfidf_query_end + host fusion + rag_flat_mmr) and (b) the oracle's brute force over the FULL
TF-IDF ranking (oracle/fusion.py, pinned to the reference by tests/golden/strategy_golden.json)."""
import numpy as np
import pytest

from oracle import fusion as ofu

pytestmark = pytest.mark.gpu


def _setup(n_docs, vocab, d, seed):
    from scipy.sparse import random as sprandom
    from sklearn.preprocessing import normalize
    from rag_system_amd import faiss_compat as faiss
    from rag_system_amd.tfidf import TfidfScorer
    rng = np.random.default_rng(seed)
    M = sprandom(n_docs, vocab, density=0.02, format="csr", random_state=seed, dtype=np.float64)
    M.data = np.round(M.data * 8) / 8 + 0.125          # few distinct values: many exact score ties
    M = normalize(M, copy=False)
    ts = TfidfScorer(None, M)
    X = rng.standard_normal((n_docs, d)).astype(np.float32)
    X[7] = X[3]
    X[11] = 0.0
    idx = faiss.IndexFlatIP(d)
    idx.add(X)
    idx.search(X[:1], 1)
    return rng, M, ts, idx, X


def _three_calls(ts, idx, tok, dense, bm25, top_k, threshold, knee, q, lam, budget, blen, flat_row):
    """The route rag_system_amd/search_engine.py took before: rag_rrf, rank query, host fusion, rag_flat_mmr."""
    from rag_system_amd.candidates import mmr_select_stored, reciprocal_rank_fusion
    from rag_system_amd.search_engine import detect_knee
    fused = reciprocal_rank_fusion([dense, bm25], k=60)
    cand = list(fused)
    rows = [c for c in cand if c >= 0]
    tI, _tD, ranks = ts.ranks_from_row_end(tok, top_k + len(cand), np.asarray(rows, dtype=np.int64))
    rk = dict(zip(rows, (int(r) for r in ranks)))
    for doc in cand:
        if doc in rk:
            fused[doc] = fused[doc] + 1.0 / (60 + rk[doc])
    for rank, doc in enumerate(int(i) for i in tI):
        if doc not in fused:
            fused[doc] = 0 + 1.0 / (60 + rank)
    order = sorted(fused.items(), key=lambda kv: kv[1], reverse=True)[:top_k]
    kept = list(order)
    if knee and kept:
        cut = detect_knee([s for _d, s in kept])
        kept = [(dd, s) for dd, s in kept if s > cut]
    if threshold is not None:
        kept = [(dd, s) for dd, s in kept if s >= threshold]
    sel = None
    if kept and all(dd >= 0 and flat_row[dd] >= 0 for dd, _s in kept):
        sel = mmr_select_stored(idx, [int(flat_row[dd]) for dd, _s in kept], q, [int(blen[dd]) for dd, _s in kept], lam,
                                top_k, budget)
    return order, kept, sel


@pytest.mark.parametrize("n_docs,d", [(5_000, 64), (300, 384), (40, 16)])
def test_hybrid_tail_equals_the_three_calls_and_the_oracle(engine, n_docs, d):
    rng, M, ts, idx, X = _setup(n_docs, 600, d, n_docs + d)
    flat_row = rng.permutation(n_docs).astype(np.int64)     # document r's embedding is row flat_row[r]
    blen = rng.integers(20, 900, n_docs).astype(np.int64)
    assert ts.set_doc_tables(flat_row, blen)
    Mc = M.tocsc()
    n_tail = n_fallback_mmr = 0
    for trial in range(40):
        top_k = int(rng.choice([1, 3, 10, 10, 20, 64]))
        nd = int(rng.integers(0, min(3 * top_k, 90, n_docs) + 1))
        nb = int(rng.integers(0, min(3 * top_k, 90) + 1))
        dense = [int(x) for x in rng.choice(n_docs, nd, replace=False)]
        pool = rng.choice(n_docs, max(1, nb // 2 + 1), replace=False)
        bm25 = [int(x) for x in rng.choice(pool, nb)]           # repeats: every entry counts
        if trial % 5 == 0 and nd > 2:
            dense[1] = -1 - 17                                      # a document without a TF-IDF row
            if nb:
                bm25[0] = -1 - 17
        nt = int(rng.integers(1, 6))
        uq = np.sort(rng.choice(600, nt, replace=False)).astype(np.int32)
        w = rng.integers(1, 4, nt).astype(np.float64)
        if trial % 7 == 0:
            w[:] = 0.0                                              # all TF-IDF scores zero: ranks = row order
        q = rng.standard_normal(d).astype(np.float32)
        lam = float(rng.choice([0.5, 0.85, 0.0, 1.0]))
        budget = None if trial % 3 else int(rng.integers(100, 3000))
        knee = bool(trial % 2)
        threshold = [None, 0.0, 0.017, 0.03][trial % 4]
        nA = len(dict.fromkeys(dense + bm25))
        kb = min(top_k + nA + 5, n_docs, 256)
        # route 1: the three calls
        tok = ts.ranks_from_row_begin(uq, w, kb)
        assert tok is not None
        order, kept, sel = _three_calls(ts, idx, tok, dense, bm25, top_k, threshold, knee, q, lam, budget, blen, flat_row)
        # route 2: one call
        tok = ts.ranks_from_row_begin(uq, w, kb)
        out = ts.hybrid_tail(tok, idx, dense, bm25, top_k, 60, threshold, knee, q, lam, budget)
        assert out is not None, "shape must be served"
        fd, fs, n_kept, mo = out
        n_tail += 1
        assert [int(x) for x in fd] == [dd for dd, _s in order], (trial, top_k)
        assert fs.tobytes() == np.array([s for _d, s in order], dtype=np.float64).tobytes()
        assert n_kept == len(kept)
        if (top_k * d + d) * 4 > (96 << 10):       # top_k rows + the query must fit the kernel's LDS image
            assert mo is None
        elif sel is None:
            assert mo is None or not kept
            n_fallback_mmr += bool(kept)
        else:
            assert mo == sel, (trial, mo, sel)
        # oracle: RRF over the FULL rankings (every document has a TF-IDF entry)
        wn = w / np.sqrt((w * w).sum()) if (w * w).sum() > 0 else w
        s_ref = np.zeros(n_docs)
        for t, ww in zip(uq, wn):
            r = Mc.indices[Mc.indptr[t]:Mc.indptr[t + 1]]
            s_ref[r] += ww * Mc.data[Mc.indptr[t]:Mc.indptr[t + 1]]
        t_order = np.lexsort((np.arange(n_docs), -s_ref))
        full = ofu.reciprocal_rank_fusion([[(i, 0.0) for i in dense], [(i, 0.0) for i in bm25],
                                           [(int(i), 0.0) for i in t_order]], k=60)
        exp = sorted(full.items(), key=lambda kv: kv[1], reverse=True)[:top_k]
        assert exp == order, trial
        if sel is not None:
            E = X[[int(flat_row[dd]) for dd, _s in kept]]
            exp_sel = ofu.mmr_selection(E, q, [int(blen[dd]) for dd, _s in kept], lam, top_k, budget)
            if exp_sel != sel:      # float32 near-tie between the GPU's and sklearn's summation order
                _s, trace = ofu.mmr_selection(E, q, [int(blen[dd]) for dd, _s in kept], lam, top_k, budget, return_trace=True)
                j = next((i for i in range(min(len(sel), len(exp_sel))) if sel[i] != exp_sel[i]), min(len(sel), len(exp_sel)))
                gaps = ofu.mmr_trace_margins(E, q, lam, trace)
                assert min(abs(g) for (ns, _p, _ok), g in zip(trace, gaps) if ns == j) < 1e-5
    assert n_tail == 40 and n_fallback_mmr >= 2


def test_hybrid_tail_limits_and_fallbacks(engine):
    rng, M, ts, idx, X = _setup(2_000, 600, 32, 5)
    uq = np.array([3, 9], dtype=np.int32)
    w = np.array([1.0, 2.0])
    dense = list(range(10))
    bm25 = list(range(5, 15))
    # no tables: fusion done, MMR left to the caller
    tok = ts.ranks_from_row_begin(uq, w, 40)
    fd, fs, n_kept, mo = ts.hybrid_tail(tok, idx, dense, bm25, 10, q=np.ones(32, np.float32))
    assert len(fd) == 10 and n_kept == 10 and mo is None
    # no flat index: fusion only
    tok = ts.ranks_from_row_begin(uq, w, 40)
    fd2, fs2, _nk, mo2 = ts.hybrid_tail(tok, None, dense, bm25, 10)
    assert fd2.tolist() == fd.tolist() and fs2.tobytes() == fs.tobytes() and mo2 is None
    # head too short for top_k + candidates: refused, the query stays begun and query_end still serves it
    tok = ts.ranks_from_row_begin(uq, w, 12)
    assert ts.hybrid_tail(tok, None, dense, bm25, 10) is None
    tI, _tD, ranks = ts.ranks_from_row_end(tok, 12, np.asarray(dense, dtype=np.int64))
    assert len(tI) == 12 and len(ranks) == 10
    # beyond the limits
    tok = ts.ranks_from_row_begin(uq, w, 200)
    assert tok is not None
    assert ts.hybrid_tail(tok, None, dense, bm25, 65) is None                       # top_k > 64
    assert ts.hybrid_tail(tok, None, list(range(129)), bm25, 10) is None            # ranking > 128 entries
    with pytest.raises(ValueError):
        ts.hybrid_tail(tok, None, [5_000_000], bm25, 10)                            # row outside the index
    # (the failed call ended the query on the Python side: a new one can begin)
    tok = ts.ranks_from_row_begin(uq, w, 40)
    assert tok is not None
    out = ts.hybrid_tail(tok, None, [], [], 10)                                     # no candidates: the TF-IDF head alone
    assert out is not None and len(out[0]) == 10
    assert out[1].tolist() == [1.0 / (60 + r) for r in range(10)]
    # a call without a begun query
    assert ts.hybrid_tail(None, None, dense, bm25, 10) is None


def _product_store(n_docs=400, dim=96, seed=3):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _data import zipf_corpus
    from rag_system_amd.text import HashingEmbeddingManager
    from rag_system_amd.vector_storage import VectorStorageManager
    em = HashingEmbeddingManager(dim=dim, chunk_size=90, overlap=10)
    store = VectorStorageManager().create_storage("s", dim, embedder=em.model_name)
    docs = zipf_corpus(n_docs, 300, 25, seed=seed)          # nearly every document is one chunk ...
    for i in range(0, n_docs, 40):                           # ... and every 40th is long enough for several
        docs[i] = list(docs[i]) * 12
    docs[9] = list(docs[4])                                  # duplicate document: exact ties in every signal
    n_single = 0
    for i, toks in enumerate(docs):
        chunks, emb = em.embed_text(" ".join(toks), doc_name="doc%d.txt" % i)
        n_single += len(chunks) == 1
        for ch, e in zip(chunks, emb):
            store.add_vector(e, "doc%d" % i, ch["chunk"], ch["source"])
    store.rebuild_bm25_index()
    store.rebuild_tfidf_index()
    assert 0.8 * n_docs < n_single < n_docs, n_single
    return store, em


def test_search_engine_rrf_through_the_one_call_tail_equals_the_three_call_route(engine):
    """SearchEngine.search_with_strategy("rrf", ...) takes rag_hybrid_rrf_mmr for everything behind the three
    rankings -- incl. knee / threshold and, with mmr_reuse_index_vectors, MMR on the stored rows of one-chunk
    documents.  Same result lists as the three-call route (the tail switched off), entry for entry, and as
    search_rrf + mmr_selection(reuse_index_vectors=True) called separately."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _data import zipf_queries
    from rag_system_amd.search_engine import SearchEngine
    store, em = _product_store()
    eng = SearchEngine(store, em)
    slow = SearchEngine(store, em)
    slow._rrf_tail = lambda *a, **k: None                   # the three-call route
    calls = {"tail": 0, "mmr_in_kernel": 0, "encode": 0}
    scorer = store.get_tfidf_scorer()
    real_tail = scorer.hybrid_tail

    def counting_tail(*a, **k):
        out = real_tail(*a, **k)
        calls["tail"] += out is not None
        calls["mmr_in_kernel"] += out is not None and out[3] is not None
        return out
    scorer.hybrid_tail = counting_tail
    real_encode = em.model.encode

    def counting_encode(texts, *a, **k):
        if not isinstance(texts, str):
            calls["encode"] += 1
        return real_encode(texts, *a, **k)
    em.model.encode = counting_encode
    queries = [" ".join(t) for t in zipf_queries(10, 300, seed=12)] + ["zzzz qqqq", store.chunks_data[3]["chunk"][:50]]
    try:
        for q in queries:
            for top_k in (3, 10):
                assert eng.search_rrf(q, top_k=top_k) == slow.search_rrf(q, top_k=top_k)
                for kw in ({}, {"use_knee_detection": True}, {"threshold": 0.02}, {"threshold": 0.018, "use_knee_detection": True},
                           {"use_mmr": True}, {"use_mmr": True, "mmr_reuse_index_vectors": True, "mmr_lambda": 0.7},
                           {"use_mmr": True, "mmr_reuse_index_vectors": True, "max_context_bytes": 900, "threshold": 0.017},
                           {"use_mmr": True, "mmr_reuse_index_vectors": True, "use_knee_detection": True}):
                    n0 = calls["tail"]
                    got = eng.search_with_strategy(q, "rrf", top_k, **kw)
                    assert calls["tail"] == n0 + 1, "the one-call tail was not taken"
                    exp = slow.search_with_strategy(q, "rrf", top_k, **kw)
                    assert got == exp, (q, top_k, kw)
    finally:
        em.model.encode = real_encode
        scorer.hybrid_tail = real_tail
    assert calls["mmr_in_kernel"] >= 10, calls         # candidate sets of one-chunk documents only: MMR in the kernel
    assert calls["encode"] >= 10, calls                # others (multi-chunk documents among them) were re-encoded
    # a cross-encoder keeps fusion + cut in the kernel and everything behind it on the host
    class FakeCE:
        def predict(self, pairs):
            return [float(len(p[1]) % 7) for p in pairs]
    a = SearchEngine(store, em, cross_encoder=FakeCE()).search_with_strategy(queries[0], "rrf", 5, use_mmr=True)
    b = SearchEngine(store, em, cross_encoder=FakeCE())
    b._rrf_tail = lambda *x, **k: None
    assert a == b.search_with_strategy(queries[0], "rrf", 5, use_mmr=True)
