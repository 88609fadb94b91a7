"""Fake store / embedding manager / cross-encoder shared by the strategy-golden generator
(tests/golden/make_strategy_golden.py, which drives the REFERENCE's SearchEngine with them in the build
container) and by the tests that replay the committed fixture (tests/golden/strategy_golden.json) through
oracle/engine_ref.py and through rag_system_amd.SearchEngine.  TESTS ONLY.

Everything here is plain data + deterministic arithmetic:
  * the store returns FIXED hit lists for `search` / `search_bm25` (the first top_k entries of the lists the
    fixture holds per query), so what the fixture pins is the strategy code above the store
    (search_engine.py:37-48, 76-135, 138-359, 362-468), not faiss / rank_bm25 arithmetic;
  * TF-IDF is sklearn's own TfidfVectorizer over the fixture's documents (importable here and on the GPU box);
  * embeddings are a pure-Python hash of the text (md5 -> fractions), identical on every machine.
"""
import copy
import hashlib

import numpy as np

DIM = 16


def hash_embed(text, dim=DIM):
    """Deterministic float32 vector for a text, NOT normalised (like SentenceTransformer.encode of a model
    without a Normalize module).  The empty text maps to the zero vector (exercises :421-423's zero-norm rule)."""
    if text == "":
        return np.zeros(dim, dtype=np.float32)
    vals = []
    i = 0
    while len(vals) < dim:
        h = hashlib.md5(("%d|%s" % (i, text)).encode("utf-8")).digest()
        for j in range(0, 16, 4):
            vals.append(int.from_bytes(h[j:j + 4], "little") / 4294967296.0 - 0.5)
        i += 1
    # a shared component so that cosines between texts are mostly positive, like real sentence embeddings
    words = text.split()
    v = np.array(vals[:dim], dtype=np.float64)
    for w in words[:64]:
        hw = hashlib.md5(("w|" + w).encode("utf-8")).digest()
        v[hw[0] % dim] += 0.75
    return v.astype(np.float32)


class FakeModel:
    def __init__(self):
        self.calls = 0

    def encode(self, texts, *a, **kw):
        if isinstance(texts, str):
            return hash_embed(texts)
        self.calls += 1
        return np.stack([hash_embed(t) for t in texts]) if len(texts) else np.zeros((0, DIM), np.float32)


class FakeEmbeddingManager:
    """embed_query as embedding_manager.py:88-98 (L2-normalised fp32, norm < 1e-9 -> 1.0)."""
    model_name = "hash-md5"
    chunk_size, overlap = 100, 20

    def __init__(self):
        self.model = FakeModel()

    def embed_query(self, query):
        v = np.array(self.model.encode(query), dtype="float32")
        n = np.linalg.norm(v)
        if n < 1e-9:
            n = 1.0
        return v / n


class FakeCrossEncoder:
    """predict(pairs) -> scores derived from the md5 of the passage.  kind: 'list' (Python floats),
    'f32' (numpy float32 array, so normalised ce_scores are numpy float32), 'const' (all equal -> the 0.5
    branch of :384-385)."""

    def __init__(self, kind="list"):
        self.kind = kind

    def predict(self, pairs):
        if self.kind == "const":
            return [1.25] * len(pairs)
        vals = []
        for _q, p in pairs:
            h = hashlib.md5(("ce|" + p).encode("utf-8")).digest()
            vals.append(int.from_bytes(h[:4], "little") / 4294967296.0 * 10.0 - 5.0)
        if self.kind == "f32":
            return np.array(vals, dtype=np.float32)
        return vals


class FakeStore:
    """The attributes / methods SearchEngine touches: indices_present, documents, search, search_bm25,
    get_tfidf_index (reference) / get_tfidf_scorer + tfidf_doc_ids (rag_system_amd)."""

    def __init__(self, inputs, indices_present=("faiss", "bm25", "tfidf"), tfidf=True, em=None):
        from sklearn.feature_extraction.text import TfidfVectorizer
        self.indices_present = list(indices_present)
        self.documents = copy.deepcopy(inputs["documents"])
        self._queries = inputs["queries"]
        em = em or FakeEmbeddingManager()
        self._by_vec = {em.embed_query(q["text"]).tobytes(): name for name, q in self._queries.items()}
        self._by_text = {q["text"]: name for name, q in self._queries.items()}
        self.tfidf_doc_ids = list(inputs["tfidf_doc_ids"])
        self.tfidf_vectorizer = self.tfidf_index = None
        self._scorer = None
        if tfidf:
            corpus = [" ".join(ch["chunk"] for ch in self.documents.get(d, {}).get("chunks", []))
                      for d in self.tfidf_doc_ids]
            self.tfidf_vectorizer = TfidfVectorizer()
            self.tfidf_index = self.tfidf_vectorizer.fit_transform(corpus)
        self._has_tfidf = tfidf

    # --- what the reference's VectorStorage offers (vector_storage.py:173-206, :223)
    def search(self, query_vector, top_k=5):
        name = self._by_vec[np.asarray(query_vector, dtype="float32").reshape(-1).tobytes()]
        return copy.deepcopy(self._queries[name]["dense_hits"][:top_k])

    def search_bm25(self, query_text, top_k=5):
        return copy.deepcopy(self._queries[self._by_text[query_text]]["bm25_hits"][:top_k])

    def get_tfidf_index(self):
        if not self._has_tfidf:
            raise RuntimeError("no TF-IDF index in this store")     # the reference catches it (:278-282)
        return self.tfidf_vectorizer, self.tfidf_index, self.tfidf_doc_ids

    # --- what rag_system_amd.SearchEngine asks for instead
    def get_tfidf_scorer(self):
        if not self._has_tfidf:
            raise RuntimeError("no TF-IDF index in this store")
        if self._scorer is None:
            from rag_system_amd.tfidf import TfidfScorer
            self._scorer = TfidfScorer(self.tfidf_vectorizer, self.tfidf_index)
        return self._scorer


def jsonable(results):
    """Result dictionaries with every number as a Python float (float32 / float64 widen exactly)."""
    out = []
    for r in results:
        o = {}
        for k, v in r.items():
            o[k] = float(v) if isinstance(v, (float, np.floating)) else v
        out.append(o)
    return out


def make_engine(cls, inputs, cfg, **extra):
    """cls(store, em, cross_encoder=...) for a fixture case configuration."""
    em = FakeEmbeddingManager()
    store = FakeStore(inputs, cfg.get("indices_present", ("faiss", "bm25", "tfidf")), cfg.get("tfidf", True), em)
    ce = FakeCrossEncoder(cfg["cross_encoder"]) if cfg.get("cross_encoder") else None
    return cls(store, em, cross_encoder=ce, **extra), store, em


def run_case(engine, inputs, case):
    """Call the engine method a fixture case names; returns the result list."""
    q = inputs["queries"][case["query"]]["text"]
    a = dict(case["args"])
    call = case["call"]
    if call == "search_with_strategy":
        return engine.search_with_strategy(q, **a)
    return getattr(engine, call)(q, **a)
