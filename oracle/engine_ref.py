"""Brute-force restatement of the reference's search strategies -- TEST INFRASTRUCTURE ONLY.

State is plain data (a float32 matrix, chunk records, a BM25OkapiRef, sklearn's vectorizer and
matrix); every strategy is computed the way the reference computes it -- every document scored,
full stable sorts, Python dicts in insertion order -- so that the product's candidate-set
shortcuts (rag_system_amd/search_engine.py) can be checked for identical output.
Follows search_engine.py:76-135 (dispatch/knee/threshold), :138-171 (faiss), :174-192 (bm25),
:195-269 (hybrid), :272-317 (multi), :320-359 (rrf); vector_storage.py:173-206 (store searches).
The reference has no tests or fixtures for these: PARITY UNPINNED (restatement only).
"""
import numpy as np

from . import dense as od
from . import fusion as ofu
from . import sparse as osp

W_DENSE, W_BM25, W_TFIDF = 0.4, 0.3, 0.3


class RefState:
    def __init__(self, X, chunks, documents, bm25, bm25_chunks, tokenizer, vectorizer=None, tfidf=None,
                 tfidf_doc_ids=(), dense_fn=None):
        # dense_fn(q (1,d), k) -> (D, I): lets a test substitute already-verified dense results
        # (e.g. the GPU's, checked against od.flat_search separately) so that every later stage
        # can be compared exactly instead of through near-tie tolerances.
        self.dense_fn = dense_fn
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
    if st.bm25 is None:
        return []
    s = st.bm25.get_scores(st.tok(text.lower()))
    return [dict(doc_id=st.bm25_chunks[i]["doc_id"], score=float(s[i]), chunk=st.bm25_chunks[i]["chunk"],
                 source=st.bm25_chunks[i]["source"]) for i in osp.stable_topk_desc(s, k)]


def faiss(st, qvec, k, mode):
    if mode != "document":
        return _desc(dense_chunks(st, qvec, k), lambda r: r["score"])[:k]
    agg = {}
    for r in dense_chunks(st, qvec, 3 * k):
        if r["doc_id"] in agg:
            agg[r["doc_id"]][0] = max(agg[r["doc_id"]][0], r["score"])
            agg[r["doc_id"]][1].append(r["chunk"])
        else:
            agg[r["doc_id"]] = [r["score"], [r["chunk"]]]
    return _desc([dict(doc_id=d, score=v[0], text=" ".join(v[1])) for d, v in agg.items()], lambda r: r["score"])[:k]


def bm25(st, text, k, mode):
    hits = bm25_chunks(st, text, k)
    if mode != "document":
        return hits
    return [dict(doc_id=r["doc_id"], score=r["score"], text=st.doc_text(r["doc_id"])) for r in hits]


def hybrid(st, qvec, text, k, mode):
    dense = dense_chunks(st, qvec, 3 * k if mode == "document" else k)
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
    return _desc(res, lambda r: r["score"])[:k]


def _tfidf_all(st, text):
    if st.vec is None or st.tfidf is None:
        return None
    return osp.tfidf_cosine(st.vec, st.tfidf, text)


def multi(st, qvec, text, k):
    dn = {r["doc_id"]: r["score"] for r in faiss(st, qvec, 3 * k, "document")}
    bn = {r["doc_id"]: r["score"] for r in bm25(st, text, 3 * k, "document")}
    s = _tfidf_all(st, text)
    tn = {} if s is None else {st.tfidf_doc_ids[i]: float(s[i]) for i in range(len(st.tfidf_doc_ids))}

    def norm(dct):
        if not dct:
            return {}
        mx = max(dct.values())
        return {a: v / mx for a, v in dct.items()} if mx > 0 else dct
    dn, bn, tn = norm(dn), norm(bn), norm(tn)
    res = [dict(doc_id=d, score=W_DENSE * dn.get(d, 0) + W_BM25 * bn.get(d, 0) + W_TFIDF * tn.get(d, 0),
                text=st.doc_text(d)) for d in dict.fromkeys(list(dn) + list(bn) + list(tn))]
    return _desc(res, lambda r: r["score"])[:k]


def rrf(st, qvec, text, k):
    r1 = [(r["doc_id"], r["score"]) for r in faiss(st, qvec, 3 * k, "document")]
    r2 = [(r["doc_id"], r["score"]) for r in bm25(st, text, 3 * k, "document")]
    s = _tfidf_all(st, text)
    r3 = [] if s is None else _desc([(st.tfidf_doc_ids[i], float(s[i])) for i in range(len(st.tfidf_doc_ids))],
                                    lambda x: x[1])
    fused = ofu.reciprocal_rank_fusion([r1, r2, r3], k=60)
    return _desc([dict(doc_id=d, score=v, text=st.doc_text(d)) for d, v in fused.items()], lambda r: r["score"])[:k]


def with_strategy(st, qvec, text, strategy, k, mode="chunk", threshold=0.0, knee=False):
    strategy = strategy.lower().strip()
    if strategy == "faiss":
        res = faiss(st, qvec, k, mode)
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
    res = _desc(res, lambda r: r["score"])
    if knee and res:
        cut = ofu.detect_knee([r["score"] for r in res])
        res = [r for r in res if r["score"] > cut]
    return [r for r in res if r["score"] >= threshold]
