#!/usr/bin/env python3
"""Generate tests/golden/strategy_golden.json by running the REFERENCE's own rag_system/search_engine.py.

BUILD-CONTAINER ONLY (needs /root/reference; never runs on the GPU box, never imported by a test).  Only the
JSON it writes -- inputs and the reference's outputs -- is committed; no reference source travels.

How the reference file is imported without its package __init__ (which needs faiss / nltk / rank_bm25,
absent here):
  * a bare module object named `rag_system` whose __path__ is /root/reference/rag_system, so that
    `rag_system.search_engine` and its `from .config import Config` resolve to the reference's files;
  * an empty module named `openai` with a dummy `ConflictError`: search_engine.py:31 imports that name and
    NEVER references it (grep: one occurrence).  It is the only stand-in; no arithmetic is stubbed --
    numpy and sklearn.metrics.pairwise.cosine_similarity are the real ones, and every function under test
    (detect_knee :37-48, search_with_strategy :76-135, search_faiss :138-171, search_bm25 :174-192,
    search_hybrid :195-269, search_multi_representation :272-317, search_rrf :320-350,
    reciprocal_rank_fusion :352-359, rerank_with_cross_encoder :362-394, mmr_selection :396-468) is the
    reference's own code object.
The store, embedding manager and cross-encoder are the fakes of tests/_strategy_fakes.py (fixed hit lists,
sklearn TF-IDF, md5 embeddings): what is pinned is the strategy layer, rows a4, a7, a10-a14 of SURVEY 8(a).

    python3 tests/golden/make_strategy_golden.py        # rewrites tests/golden/strategy_golden.json
"""
import contextlib
import importlib
import io
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference"

import numpy as np  # noqa: E402

import _strategy_fakes as fk  # noqa: E402


def import_reference_search_engine():
    pkg = types.ModuleType("rag_system")
    pkg.__path__ = [os.path.join(REF, "rag_system")]
    sys.modules["rag_system"] = pkg
    stub = types.ModuleType("openai")

    class ConflictError(Exception):     # imported by search_engine.py:31, never used
        pass
    stub.ConflictError = ConflictError
    sys.modules["openai"] = stub
    mod = importlib.import_module("rag_system.search_engine")
    assert os.path.realpath(mod.__file__) == os.path.realpath(os.path.join(REF, "rag_system", "search_engine.py"))
    return mod


# ----------------------------------------------------------------------------- inputs

VOCAB = ("alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho sigma tau "
         "upsilon phi chi psi omega river stone cloud ember frost grove harbor island jungle kernel lantern "
         "meadow nebula orchid prairie quartz ridge summit tundra valley willow zenith anchor breeze canyon "
         "delta").split()


def build_inputs(seed=20251111):
    rng = random.Random(seed)
    weights = [1.0 / (i + 1) ** 0.9 for i in range(len(VOCAB))]
    documents, order = {}, []
    for i in range(44):
        doc_id = "doc%02d" % i
        chunks = []
        for c in range(rng.randint(1, 4)):
            words = rng.choices(VOCAB, weights, k=rng.randint(5, 14))
            chunks.append({"chunk": " ".join(words), "source": "%s.txt#%d" % (doc_id, c)})
        documents[doc_id] = {"chunks": chunks}
        order.append(doc_id)
    # exact duplicates (ties in every signal), a one-chunk copy of another document's first chunk
    documents["doc07"] = {"chunks": [dict(c) for c in documents["doc03"]["chunks"]]}
    documents["doc21"] = {"chunks": [dict(documents["doc12"]["chunks"][0])]}
    documents["doc30"] = {"chunks": [{"chunk": "", "source": None}]}       # empty text: zero TF-IDF row, zero vector
    chunks_flat = [(d, ch["chunk"], ch["source"]) for d in order for ch in documents[d]["chunks"]]

    queries_text = {
        "q_common": "alpha beta gamma",
        "q_mid": "river stone cloud ember frost",
        "q_rare": "zenith anchor willow",
        "q_rep": "delta delta epsilon alpha",
        "q_oov": "zzzz qqqq",                                   # no vocabulary overlap: all TF-IDF scores zero
        "q_chunk": documents["doc05"]["chunks"][0]["chunk"],    # a stored chunk as the query
        "q_mixed": "kappa zzzz omega grove",
    }
    em = fk.FakeEmbeddingManager()
    E = np.stack([fk.hash_embed(c[1]) for c in chunks_flat]).astype(np.float32)
    nrm = np.linalg.norm(E, axis=1, keepdims=True)
    nrm[nrm == 0] = 1e-9
    E = (E / nrm).astype(np.float32)
    n = len(chunks_flat)
    df = {}
    toks = [c[1].split() for c in chunks_flat]
    for t in toks:
        for w in set(t):
            df[w] = df.get(w, 0) + 1
    queries = {}
    for name, text in queries_text.items():
        qv = em.embed_query(text)
        s = (E @ qv).astype(np.float32)
        idx = sorted(range(n), key=lambda i: (-float(s[i]), i))[:90]
        dense = [{"doc_id": chunks_flat[i][0], "score": float(s[i]), "chunk": chunks_flat[i][1],
                  "source": chunks_flat[i][2]} for i in idx]
        # fixed sparse hit list in the shape vector_storage.search_bm25 returns (:173-190): float64 scores,
        # stable descending order over ALL chunks (zero scores included), doc ids may repeat
        sc = []
        for i in range(n):
            v = 0.0
            for w in text.lower().split():
                tf = toks[i].count(w)
                if tf:
                    v += np.log(1.0 + n / df[w]) * (tf * 2.0 / (tf + 1.0))
            sc.append(float(v))
        bidx = sorted(range(n), key=lambda i: sc[i], reverse=True)[:30]
        sparse = [{"doc_id": chunks_flat[i][0], "score": sc[i], "chunk": chunks_flat[i][1],
                   "source": chunks_flat[i][2]} for i in bidx]
        queries[name] = {"text": text, "dense_hits": dense, "bm25_hits": sparse}
    # a BM25 hit whose document the store does not know (documents.get(doc_id, {}) -> empty text)
    queries["q_mid"]["bm25_hits"][2]["doc_id"] = "ghost"
    # planted exact score ties between different documents in the dense list
    queries["q_rare"]["dense_hits"][4]["score"] = queries["q_rare"]["dense_hits"][3]["score"]
    return {"dim": fk.DIM, "documents": documents, "tfidf_doc_ids": order, "queries": queries}


# ----------------------------------------------------------------------------- cases

def strategy_cases(inputs):
    cases = []
    qnames = list(inputs["queries"])
    full = {"indices_present": ["faiss", "bm25", "tfidf"], "tfidf": True, "cross_encoder": None}

    def add(call, query, args, cfg=None):
        c = dict(full)
        c.update(cfg or {})
        cases.append({"call": call, "query": query, "args": args, "engine": c})

    for q in qnames:
        for k in (1, 3, 10):
            for mode in ("chunk", "document"):
                add("search_faiss", q, {"top_k": k, "retrieval_mode": mode})
                add("search_bm25", q, {"top_k": k, "retrieval_mode": mode})
                add("search_hybrid", q, {"top_k": k, "retrieval_mode": mode})
            add("search_multi_representation", q, {"top_k": k})
            add("search_rrf", q, {"top_k": k})
        for strategy, mode in (("faiss", "chunk"), ("FAISS ", "document"), ("bm25", "document"), ("ibrido", "chunk"),
                               ("hybrid", "document"), ("multi", "chunk"), ("rrf", "chunk")):
            add("search_with_strategy", q, {"strategy": strategy, "top_k": 10, "retrieval_mode": mode})
            add("search_with_strategy", q, {"strategy": strategy, "top_k": 10, "retrieval_mode": mode,
                                            "use_knee_detection": True})
            add("search_with_strategy", q, {"strategy": strategy, "top_k": 5, "retrieval_mode": mode,
                                            "threshold": 0.05, "use_knee_detection": True})
            add("search_with_strategy", q, {"strategy": strategy, "top_k": 10, "retrieval_mode": mode,
                                            "threshold": 0.3})
    # missing indices: each signal switched off in turn (indices_present / no TF-IDF index)
    for q in ("q_common", "q_mid", "q_oov"):
        for cfg in ({"indices_present": ["bm25", "tfidf"]}, {"indices_present": ["faiss", "tfidf"]},
                    {"indices_present": ["faiss", "bm25"], "tfidf": False}, {"indices_present": [], "tfidf": False}):
            for call, args in (("search_faiss", {"top_k": 5, "retrieval_mode": "document"}),
                               ("search_bm25", {"top_k": 5, "retrieval_mode": "document"}),
                               ("search_hybrid", {"top_k": 5, "retrieval_mode": "document"}),
                               ("search_hybrid", {"top_k": 5, "retrieval_mode": "chunk"}),
                               ("search_multi_representation", {"top_k": 5}),
                               ("search_rrf", {"top_k": 5})):
                add(call, q, args, cfg)
    # cross-encoder hook (a fake model: the hook's arithmetic is what is pinned)
    for kind in ("list", "f32", "const"):
        for q in ("q_common", "q_rare", "q_chunk"):
            for strategy, mode in (("faiss", "chunk"), ("faiss", "document"), ("bm25", "chunk"), ("bm25", "document"),
                                   ("ibrido", "chunk"), ("ibrido", "document"), ("multi", "document"), ("rrf", "document")):
                add("search_with_strategy", q, {"strategy": strategy, "top_k": 6, "retrieval_mode": mode},
                    {"cross_encoder": kind})
            add("search_with_strategy", q, {"strategy": "faiss", "top_k": 8, "retrieval_mode": "chunk",
                                            "threshold": 0.4, "use_knee_detection": True}, {"cross_encoder": kind})
    # MMR behind a strategy (candidates re-encoded by the fake model), with and without a byte budget
    for q in ("q_common", "q_mid", "q_rare", "q_chunk", "q_mixed"):
        for strategy, mode in (("faiss", "chunk"), ("faiss", "document"), ("bm25", "chunk"), ("ibrido", "chunk"),
                               ("multi", "document"), ("rrf", "document")):
            for lam, budget in ((0.5, None), (0.85, None), (0.3, 160), (1.0, 60)):
                add("search_with_strategy", q, {"strategy": strategy, "top_k": 8, "retrieval_mode": mode,
                                                "threshold": -1.0, "use_mmr": True, "mmr_lambda": lam,
                                                "max_context_bytes": budget})
    return cases


def direct_cases(inputs):
    rng = random.Random(99)
    knee = [[], [0.7], [0.9, 0.9], [0.9, 0.5, 0.45, 0.1], [1.0, 0.9, 0.8, 0.7], [0.5, 0.5, 0.2, 0.2, -0.1],
            [0.3, 0.6, 0.1], [3.0, 2.0, 1.0, 0.0, 0.0], [float(np.float32(0.7)), float(np.float32(0.1)), 0.05]]
    knee += [sorted((rng.random() for _ in range(rng.randint(2, 12))), reverse=True) for _ in range(12)]

    def ranking(n, ids=12, ties=False):
        out = []
        for _ in range(n):
            s = rng.choice([0.0, 0.25, 0.5, 1.0]) if ties else rng.random() * 3
            out.append({"doc_id": "d%d" % rng.randrange(ids), "score": s})
        return out
    rrf = [{"k": 60, "rankings": [[], [], []]},
           {"k": 60, "rankings": [ranking(5), [], ranking(7)]},
           {"k": 60, "rankings": [ranking(9, ties=True), ranking(9, ties=True), ranking(30, ids=40, ties=True)]},
           {"k": 10, "rankings": [ranking(12, ids=5), ranking(3, ids=5)]},
           {"k": 60, "rankings": [ranking(40, ids=25), ranking(40, ids=25, ties=True), ranking(80, ids=100),
                                  ranking(6)]},
           {"k": 1, "rankings": [ranking(20, ids=8, ties=True)]}]
    texts = [ch["chunk"] for d in inputs["documents"].values() for ch in d["chunks"]]
    ce = []
    for kind in ("list", "f32", "const"):
        for normalize in (True, False):
            for n in (0, 1, 2, 7):
                res = [({"doc_id": "d%d" % i, "score": rng.random(), "chunk": rng.choice(texts)} if i % 2 else
                        {"doc_id": "d%d" % i, "score": rng.random(), "text": rng.choice(texts)}) for i in range(n)]
                if n == 7:
                    res[3] = {"doc_id": "d3", "score": 0.1}            # neither text nor chunk -> ""
                ce.append({"kind": kind, "normalize_ce": normalize, "query": "alpha beta", "results": res})
    mmr = []
    for n, lam, top_k, budget in ((1, 0.5, None, None), (2, 0.5, None, None), (6, 0.5, 3, None), (12, 0.85, 10, None),
                                  (12, 0.0, None, None), (12, 1.0, 5, None), (20, 0.5, 10, 120), (20, 0.3, 20, 400),
                                  (9, 0.7, 9, 1), (30, 0.6, 10, None), (30, 0.6, None, 300), (15, 0.5, 40, None)):
        res = []
        for i in range(n):
            t = rng.choice(texts)
            res.append({"doc_id": "m%d" % i, "score": rng.random(), ("chunk" if i % 3 else "text"): t})
        if n >= 12:
            res[5] = dict(res[2], doc_id="m5")          # identical texts: cosine 1 between candidates, tied relevance
            res[7] = {"doc_id": "m7", "score": 0.2, "chunk": ""}     # zero vector -> norm replaced by 1e-9
        mmr.append({"query": rng.choice(list(inputs["queries"])), "lambda_param": lam, "top_k": top_k,
                    "max_context_bytes": budget, "results": res})
    return {"detect_knee": [{"scores": s} for s in knee], "reciprocal_rank_fusion": rrf,
            "rerank_with_cross_encoder": ce, "mmr_selection": mmr}


# ----------------------------------------------------------------------------- run the reference

def main():
    ref = import_reference_search_engine()
    inputs = build_inputs()
    quiet = io.StringIO()       # the reference prints [DEBUG] lines
    cases = strategy_cases(inputs)
    with contextlib.redirect_stdout(quiet):
        for c in cases:
            eng, _store, _em = fk.make_engine(ref.SearchEngine, inputs, c["engine"])
            c["expected"] = fk.jsonable(fk.run_case(eng, inputs, c))
            if c["args"].get("use_mmr"):        # the candidate list MMR was given (same call without MMR)
                eng, _store, _em = fk.make_engine(ref.SearchEngine, inputs, c["engine"])
                c["expected_pre_mmr"] = fk.jsonable(fk.run_case(eng, inputs, dict(c, args=dict(c["args"], use_mmr=False))))
        direct = direct_cases(inputs)
        for c in direct["detect_knee"]:
            c["expected"] = float(ref.detect_knee(list(c["scores"])))
        eng, _store, em = fk.make_engine(ref.SearchEngine, inputs, {})
        for c in direct["reciprocal_rank_fusion"]:
            got = eng.reciprocal_rank_fusion(*[[dict(r) for r in rk] for rk in c["rankings"]], k=c["k"])
            c["expected"] = [[d, float(s)] for d, s in got.items()]
        for c in direct["rerank_with_cross_encoder"]:
            e2, _s, _e = fk.make_engine(ref.SearchEngine, inputs, {"cross_encoder": c["kind"]})
            got = e2.rerank_with_cross_encoder(c["query"], [dict(r) for r in c["results"]], normalize_ce=c["normalize_ce"])
            c["expected"] = fk.jsonable(got)
        for c in direct["mmr_selection"]:
            res = [dict(r) for r in c["results"]]
            got = eng.mmr_selection(inputs["queries"][c["query"]]["text"], res, em, lambda_param=c["lambda_param"],
                                    top_k=c["top_k"], max_context_bytes=c["max_context_bytes"])
            c["expected_doc_ids"] = [r["doc_id"] for r in got]
    out = {"generator": "tests/golden/make_strategy_golden.py",
           "reference": "rag_system/search_engine.py (imported from /root/reference; numpy %s, scikit-learn %s)"
                        % (np.__version__, __import__("sklearn").__version__),
           "inputs": inputs, "cases": cases, "direct": direct}
    path = os.path.join(HERE, "strategy_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    n_res = sum(len(c["expected"]) for c in cases)
    print("wrote %s: %d strategy cases (%d result entries), %d direct cases, %.1f KiB"
          % (path, len(cases), n_res, sum(len(v) for v in direct.values()), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
