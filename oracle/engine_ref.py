"""Brute-force restatement of the reference's search strategies -- TEST INFRASTRUCTURE ONLY.

State is plain data (a float32 matrix, chunk records, a BM25OkapiRef, sklearn's vectorizer and
matrix); every strategy is computed the way the reference computes it -- every document scored,
full stable sorts, Python dicts in insertion order -- so that the product's candidate-set
shortcuts (rag_system_amd/search_engine.py) can be checked for identical output.
Follows search_engine.py:76-135 (dispatch/knee/threshold), :138-171 (faiss), :174-192 (bm25),
:195-269 (hybrid), :272-317 (multi), :320-359 (rrf); vector_storage.py:173-206 (store searches).
PINNED (round 3) by tests/golden/strategy_golden.json: the outputs of the reference's own
rag_system/search_engine.py, imported in the build container by tests/golden/make_strategy_golden.py and
driven with a fake store (fixed dense / BM25 hit lists, sklearn TF-IDF); tests/test_strategy_golden.py
replays every case through this file.  The dense and BM25 SCORES below the store (faiss, rank_bm25) stay
unpinned -- see oracle/dense.py and oracle/sparse.py.
"""
import numpy as np

from . import dense as od
from . import fusion as ofu
from . import sparse as osp

W_DENSE, W_BM25, W_TFIDF = 0.4, 0.3, 0.3


class RefState:
    def __init__(self, X, chunks, documents, bm25, bm25_chunks, tokenizer, vectorizer=None, tfidf=None,
                 tfidf_doc_ids=(), dense_fn=None, dense_hits_fn=None, bm25_hits_fn=None,
                 indices_present=("faiss", "bm25", "tfidf"), cross_encoder=None, encode_fn=None):
        # dense_fn(q (1,d), k) -> (D, I): lets a test substitute already-verified dense results
        # (e.g. the GPU's, checked against od.flat_search separately) so that every later stage
        # can be compared exactly instead of through near-tie tolerances.
        self.dense_fn = dense_fn
        # dense_hits_fn(qvec, k) / bm25_hits_fn(text, k) -> the store's hit lists themselves (what
        # vector_storage.search / search_bm25 return): the strategy-golden replay feeds the fixture's lists.
        self.dense_hits_fn, self.bm25_hits_fn = dense_hits_fn, bm25_hits_fn
        self.indices_present = list(indices_present)
        self.cross_encoder = cross_encoder      # any object with predict(pairs) (search_engine.py:56)
        self.encode_fn = encode_fn              # texts -> embeddings, for MMR (:418)
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        self.chunks = chunks              # [{doc_id, chunk, source}] aligned with X rows
        self.documents = documents        # doc_id -> {'chunks': [{'chunk','source'}]}
        self.bm25, self.bm25_chunks, self.tok = bm25, bm25_chunks, tokenizer
        self.vec, self.tfidf, self.tfidf_doc_ids = vectorizer, tfidf, list(tfidf_doc_ids)

    def doc_text(self, d):
        return " ".join(c["chunk"] for c in self.documents.get(d, {}).get("chunks", []))


def _desc(items, key):
    return sorted(items, key=key, reverse=True)


def dense_chunks(st, qvec, k):
    if st.dense_hits_fn is not None:
        hits = st.dense_hits_fn(qvec, k)
        for r in hits:
            r["score"] = float(r["score"])
        return hits
    if len(st.X) == 0:
        return []
    q = np.asarray(qvec, np.float32).reshape(1, -1)
    if st.dense_fn is not None:
        D, I = st.dense_fn(q, k)
    else:
        D, I, _ = od.flat_search(st.X, q, k, od.METRIC_IP, accum64=True)
    return [dict(doc_id=st.chunks[i]["doc_id"], score=float(s), chunk=st.chunks[i]["chunk"],
                 source=st.chunks[i]["source"]) for s, i in zip(D[0], I[0])]


def bm25_chunks(st, text, k):
    if st.bm25_hits_fn is not None:
        return st.bm25_hits_fn(text, k)
    if st.bm25 is None:
        return []
    s = st.bm25.get_scores(st.tok(text.lower()))
    return [dict(doc_id=st.bm25_chunks[i]["doc_id"], score=float(s[i]), chunk=st.bm25_chunks[i]["chunk"],
                 source=st.bm25_chunks[i]["source"]) for i in osp.stable_topk_desc(s, k)]


def rerank(st, text, results, normalize_ce=True):
    """rerank_with_cross_encoder, search_engine.py:362-394: ce_score = model score, min-max normalised
    (all within 1e-6 of each other -> 0.5) unless normalize_ce is False; stable sort by ce_score."""
    if not results:
        return results
    ce = np.array(st.cross_encoder.predict([(text, r.get("text", r.get("chunk", ""))) for r in results]))
    if normalize_ce:
        lo, hi = (float(ce.min()), float(ce.max())) if ce.size > 0 else (0.0, 0.0)
        vals = [(v - lo) / (hi - lo) for v in ce] if hi - lo > 1e-6 else [0.5] * len(ce)
    else:
        vals = [float(v) for v in ce]
    for r, v in zip(results, vals):
        r["ce_score"] = v
    results.sort(key=lambda r: r["ce_score"], reverse=True)
    return results


def faiss(st, qvec, text, k, mode):
    if "faiss" not in st.indices_present:
        return []
    if mode != "document":
        res = dense_chunks(st, qvec, k)
    else:
        agg = {}
        for r in dense_chunks(st, qvec, 3 * k):
            if r["doc_id"] in agg:
                agg[r["doc_id"]][0] = max(agg[r["doc_id"]][0], r["score"])
                agg[r["doc_id"]][1].append(r["chunk"])
            else:
                agg[r["doc_id"]] = [r["score"], [r["chunk"]]]
        res = [dict(doc_id=d, score=v[0], text=" ".join(v[1])) for d, v in agg.items()]
    if st.cross_encoder:
        res = rerank(st, text, res, True)
    return _desc(res, lambda r: r["score"])[:k]          # by score again, even after a rerank (:162, :170)


def bm25(st, text, k, mode):
    if "bm25" not in st.indices_present:
        return []
    hits = bm25_chunks(st, text, k)
    if mode != "document":
        return rerank(st, text, hits, True) if st.cross_encoder else hits
    docs = [dict(doc_id=r["doc_id"], score=r["score"], text=st.doc_text(r["doc_id"])) for r in hits]
    return rerank(st, text, docs, False) if st.cross_encoder else docs


def hybrid(st, qvec, text, k, mode):
    dense = dense_chunks(st, qvec, 3 * k if mode == "document" else k) if "faiss" in st.indices_present else []
    sparse = bm25(st, text, k, mode)
    out = {}
    if mode == "document":
        grp = {}
        for r in dense:
            g = grp.setdefault(r["doc_id"], [r["score"], []])
            g[0] = max(g[0], r["score"])
            g[1].append(r["chunk"])
        for d, g in grp.items():
            out[d] = W_DENSE * g[0]
        for r in sparse:
            out[r["doc_id"]] = out.get(r["doc_id"], 0) + W_BM25 * r["score"]
        res = [dict(doc_id=d, score=s, text=" ".join(grp[d][1]) if d in grp else st.doc_text(d), source="hybrid")
               for d, s in out.items()]
    else:
        best = {}
        for r in dense:
            if r["doc_id"] not in best or r["score"] > best[r["doc_id"]][1]:
                best[r["doc_id"]] = (r["chunk"], r["score"])
        for d, (_c, s) in best.items():
            out[d] = W_DENSE * s
        for r in sparse:
            if r["doc_id"] in best:
                out[r["doc_id"]] = out.get(r["doc_id"], 0) + W_BM25 * r["score"]
        res = [dict(doc_id=d, score=s, chunk=best[d][0], source="hybrid") for d, s in out.items()]
    res = _desc(res, lambda r: r["score"])[:k]
    return rerank(st, text, res, True) if st.cross_encoder else res


def _tfidf_all(st, text):
    if st.vec is None or st.tfidf is None:
        return None
    return osp.tfidf_cosine(st.vec, st.tfidf, text)


def multi(st, qvec, text, k, cut=True):
    dn = {r["doc_id"]: r["score"] for r in faiss(st, qvec, text, 3 * k, "document")}
    bn = {r["doc_id"]: r["score"] for r in bm25(st, text, 3 * k, "document")}      # later duplicates overwrite (:276)
    s = _tfidf_all(st, text)
    tn = {} if s is None else {st.tfidf_doc_ids[i]: float(s[i]) for i in range(len(st.tfidf_doc_ids))}

    def norm(dct):
        if not dct:
            return {}
        mx = max(dct.values())
        return {a: v / mx for a, v in dct.items()} if mx > 0 else dct
    dn, bn, tn = norm(dn), norm(bn), norm(tn)
    # the reference iterates a SET of doc ids (:301-304), whose order is not defined; what is defined is the
    # result after the stable sort for distinct scores.  Equal fused scores are compared as sets by the tests.
    res = [dict(doc_id=d, score=W_DENSE * dn.get(d, 0) + W_BM25 * bn.get(d, 0) + W_TFIDF * tn.get(d, 0),
                text=st.doc_text(d)) for d in dict.fromkeys(list(dn) + list(bn) + list(tn))]
    res = _desc(res, lambda r: r["score"])
    if not cut:
        return res          # tests: is there a tie across the top_k boundary?
    res = res[:k]
    return rerank(st, text, res, True) if st.cross_encoder else res


def rrf(st, qvec, text, k):
    r1 = [(r["doc_id"], r["score"]) for r in faiss(st, qvec, text, 3 * k, "document")]
    r2 = [(r["doc_id"], r["score"]) for r in bm25(st, text, 3 * k, "document")]
    s = _tfidf_all(st, text)
    r3 = [] if s is None else _desc([(st.tfidf_doc_ids[i], float(s[i])) for i in range(len(st.tfidf_doc_ids))],
                                    lambda x: x[1])
    fused = ofu.reciprocal_rank_fusion([r1, r2, r3], k=60)
    res = _desc([dict(doc_id=d, score=v, text=st.doc_text(d)) for d, v in fused.items()], lambda r: r["score"])[:k]
    return rerank(st, text, res, True) if st.cross_encoder else res


def _key(r):
    return r.get("ce_score", r.get("score", 0.0))


def mmr(st, qvec, results, lam, top_k, max_context_bytes):
    """mmr_selection, search_engine.py:396-468: candidates' texts ('chunk' else 'text') re-encoded by the
    model, then the greedy loop of oracle/fusion.py."""
    if not results:
        return results
    texts = [r.get("chunk", r.get("text", "")) for r in results]
    order = ofu.mmr_selection(st.encode_fn(texts), qvec, [len(t.encode("utf-8")) for t in texts], lam,
                              top_k if top_k is not None else len(results), max_context_bytes)
    return [results[i] for i in order]


def with_strategy(st, qvec, text, strategy, k, mode="chunk", threshold=0.0, knee=False, use_mmr=False,
                  mmr_lambda=0.5, max_context_bytes=None):
    strategy = strategy.lower().strip()
    mode = mode.lower().strip()
    if strategy == "faiss":
        res = faiss(st, qvec, text, k, mode)
    elif strategy == "bm25":
        res = bm25(st, text, k, mode)
    elif strategy in ("ibrido", "hybrid"):
        res = hybrid(st, qvec, text, k, mode)
    elif strategy == "multi":
        res = multi(st, qvec, text, k)
    elif strategy == "rrf":
        res = rrf(st, qvec, text, k)
    else:
        raise ValueError("Unknown retrieval strategy: %s" % strategy)
    res = _desc(res, _key)
    if knee and res:
        cut = ofu.detect_knee([_key(r) for r in res])
        res = [r for r in res if _key(r) > cut]
    res = [r for r in res if _key(r) >= threshold]
    if use_mmr and res:
        res = mmr(st, qvec, res, mmr_lambda, k, max_context_bytes)
    return res
