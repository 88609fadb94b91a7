"""VectorStorage / VectorStorageManager with the reference's surface, indices in HBM.

Mirrors rag_system/vector_storage.py (class VectorStorage :58, VectorStorageManager :314,
align_embedding_manager :27): same constructor, attributes, method names, result dictionaries
and on-disk layout (index.faiss, index_bm25.pkl, tfidf.pkl, metadata.pkl with the same keys),
so callers written against the reference (its UI, examples/*.py, SearchEngine) keep working.

What differs is where the numbers are computed:
  self.index   faiss_compat.IndexFlatIP  -> rag_flat_* (csrc/flat.hip, flat_mfma.hip)
  self.bm25    bm25.BM25Okapi            -> rag_bm25_scores (csrc/sparse.hip)
  TF-IDF       sklearn builds the matrix exactly as the reference does (:216-217, so tfidf.pkl
               stays interchangeable); per-query scoring goes through tfidf.TfidfScorer
               -> rag_tfidf_scores.  The scorer is an HBM cache of self.tfidf_index and is
               rebuilt whenever that attribute is replaced.

Reference behaviours kept on purpose: index.search pads with id -1 when
k > ntotal and chunks_data[-1] is then the LAST chunk (:196-205); save() re-encodes every
chunk without L2 normalisation (:148-152); load() rebuilds BM25 with the library defaults
k1=1.5, b=0.75 (:305) while rebuild_bm25_index uses Config's 1.0 / 0.65 (:170);
remove_document never shrinks the BM25 corpus (:119-135,:160).
"""
import os
import pickle
import shutil

import numpy as np

from . import faiss_compat as faiss
from . import safe_pickle
from . import text as _text
from .bm25 import BM25Okapi
from .config import Config
from .text import word_tokenize
from .tfidf import TfidfScorer

_FILES = {"faiss": "index.faiss", "bm25": "index_bm25.pkl", "tfidf": "tfidf.pkl", "meta": "metadata.pkl"}


def align_embedding_manager(embedding_manager, store, manager_factory=None):
    """Force the manager's chunking parameters / embedder to the store's (vector_storage.py:27-56).
    manager_factory(model_key, chunk_size, overlap) is used to reload a different embedder."""
    changed = False
    for attr in ("chunk_size", "overlap"):
        if getattr(embedding_manager, attr) != getattr(store, attr):
            print("WARNING: %s mismatch: EmbeddingManager has %s but store requires %s. Forcing alignment."
                  % (attr, getattr(embedding_manager, attr), getattr(store, attr)))
            setattr(embedding_manager, attr, getattr(store, attr))
            changed = True
    if embedding_manager.model_name != store.embedder:
        print("WARNING: embedder mismatch: EmbeddingManager has %s but store requires %s."
              % (embedding_manager.model_name, store.embedder))
        if manager_factory is not None:
            fresh = manager_factory(store.embedder, store.chunk_size, store.overlap)
            embedding_manager.model = fresh.model
            embedding_manager.model_name = fresh.model_name
            changed = True
    if changed:
        print("EmbeddingManager parameters have been aligned to the store's parameters.")
    return embedding_manager


class VectorStorage:
    def __init__(self, embedding_dim, embedder="all-MiniLM-L6-v2", indices_present=None, chunk_size=100,
                 overlap=20):
        self.embedding_dim = embedding_dim
        self.embedder = embedder
        self.indices_present = ["faiss", "bm25", "tfidf"] if indices_present is None else indices_present
        self.chunk_size = chunk_size
        self.overlap = overlap
        self.index = faiss.IndexFlatIP(embedding_dim)
        self.bm25_corpus = []
        self.bm25_doc_ids = []
        self.bm25 = None
        self.bm25_indexed_count = 0
        self.tfidf_index = None
        self.tfidf_vectorizer = None
        self.tfidf_doc_ids = []
        self.documents = {}    # doc_id -> {'chunks': [{'chunk', 'source'}]}
        self.signatures = {}   # md5 -> doc_id
        self.doc_ids = []
        self.chunks_data = []
        self._tfidf_scorer = None
        self._tfidf_scorer_key = None
        self._dense_prefilter = False

    def set_dense_prefilter(self, enable=True):
        """Not in the reference, off by default: the dense index keeps a bfloat16 copy of its rows as candidate
        source for searches with top_k <= 26 (faiss_compat.IndexFlat.set_prefilter / rag_flat_prefilter).  The
        hits are those of the fp32 rows; +50 % HBM for the index.  Survives rebuild_faiss_index."""
        self._dense_prefilter = bool(enable)
        if hasattr(self.index, "set_prefilter"):
            self.index.set_prefilter(self._dense_prefilter)

    def set_chunk_params(self, chunk_size, overlap):
        self.chunk_size = chunk_size
        self.overlap = overlap

    # ---- dense ------------------------------------------------------------------------

    def add_vector(self, vector, doc_id, chunk_text, source=None):
        row = np.array(vector, dtype="float32").reshape(1, -1)
        self.index.add(row)   # staged on the host, flushed to HBM in bulk (rag_flat_add)
        self.doc_ids.append(doc_id)
        self.chunks_data.append({"doc_id": doc_id, "chunk": chunk_text, "source": source})
        self.documents.setdefault(doc_id, {"chunks": []})["chunks"].append({"chunk": chunk_text, "source": source})

    def search(self, query_vector, top_k=5):
        if self.index.ntotal == 0:
            return []
        q = np.array(query_vector, dtype="float32").reshape(1, -1)
        distances, indices = self.index.search(q, top_k)
        hits = []
        for dist, idx in zip(distances[0], indices[0]):
            cdata = self.chunks_data[idx]   # idx == -1 (k > ntotal) is the last chunk, as in the reference
            hits.append({"doc_id": cdata["doc_id"], "score": float(dist), "chunk": cdata["chunk"],
                         "source": cdata["source"]})
        return hits

    def search_begin(self, query_vector, top_k=5):
        """search() in two halves (not in the reference): the dense pass is enqueued and runs while the
        caller computes something else; search_end(token) returns what search() would have."""
        if self.index.ntotal == 0 or not hasattr(self.index, "search_begin"):
            return ("done", self.search(query_vector, top_k))
        q = np.array(query_vector, dtype="float32").reshape(1, -1)
        return ("pending", self.index.search_begin(q, top_k))

    def search_end(self, token, ids_only=False):
        """ids_only=True (not in the reference; the fusion strategies, which read nothing but doc_id and score of
        the dense hits): [(doc_id, score)] in hit order instead of the result dictionaries."""
        kind, val = token
        if kind == "done":
            return [(r["doc_id"], float(r["score"])) for r in val] if ids_only else val
        distances, indices = self.index.search_end(val)
        if ids_only:
            doc_ids = self.doc_ids          # aligned with chunks_data; -1 (k > ntotal) is the last chunk there too
            return [(doc_ids[i], s) for s, i in zip(distances[0].tolist(), indices[0].tolist())]
        hits = []
        for dist, idx in zip(distances[0], indices[0]):
            cdata = self.chunks_data[idx]   # idx == -1 (k > ntotal) is the last chunk, as in the reference
            hits.append({"doc_id": cdata["doc_id"], "score": float(dist), "chunk": cdata["chunk"],
                         "source": cdata["source"]})
        return hits

    def search_batch(self, query_vectors, top_k=5):
        """Many queries in one index.search call (nq x d) -- the batched front end the reference
        lacks; per query the same list search() returns."""
        if self.index.ntotal == 0:
            return [[] for _ in range(len(query_vectors))]
        q = np.ascontiguousarray(query_vectors, dtype="float32").reshape(len(query_vectors), -1)
        distances, indices = self.index.search(q, top_k)
        out = []
        for drow, irow in zip(distances, indices):
            out.append([{"doc_id": self.chunks_data[i]["doc_id"], "score": float(s),
                         "chunk": self.chunks_data[i]["chunk"], "source": self.chunks_data[i]["source"]}
                        for s, i in zip(drow, irow)])
        return out

    def rebuild_faiss_index(self, embedding_manager):
        self.index = faiss.IndexFlatIP(self.embedding_dim)
        if self._dense_prefilter:
            self.index.set_prefilter(True)
        if not self.chunks_data:
            return
        texts = [cd["chunk"] for cd in self.chunks_data]
        emb = np.array(embedding_manager.model.encode(texts), dtype="float32")   # raw model outputs (:149-152)
        self.index.add(emb)

    # ---- BM25 -------------------------------------------------------------------------

    def rebuild_bm25_index(self):
        """Tokenise the chunks added since the last call and rebuild BM25Okapi over the whole
        corpus with Config's k1 / b (vector_storage.py:154-171)."""
        new_chunks = self.chunks_data[self.bm25_indexed_count:]
        if not new_chunks:
            return
        new_tokens = [word_tokenize(cd["chunk"].lower()) for cd in new_chunks]
        if self.bm25 is None:
            self.bm25_corpus = new_tokens
            self.bm25_doc_ids = new_chunks
        else:
            self.bm25_corpus.extend(new_tokens)
            self.bm25_doc_ids.extend(new_chunks)
        self.bm25 = BM25Okapi(self.bm25_corpus, k1=Config.BM25_PARAM_K1, b=Config.BM25_PARAM_B)
        self.bm25_indexed_count = len(self.chunks_data)

    def search_bm25(self, query_text, top_k=5):
        if not self.bm25:
            return []
        tokens = word_tokenize(query_text.lower())
        ids, scores = self.bm25.top_k(tokens, top_k)   # = sorted(range(n), key=scores, reverse=True)[:top_k]
        hits = []
        for idx, sc in zip(ids, scores):
            cdata = self.bm25_doc_ids[int(idx)]
            hits.append({"doc_id": cdata["doc_id"], "score": float(sc), "chunk": cdata["chunk"],
                         "source": cdata["source"]})
        return hits

    # ---- TF-IDF -----------------------------------------------------------------------

    def rebuild_tfidf_index(self):
        from sklearn.feature_extraction.text import TfidfVectorizer
        corpus, self.tfidf_doc_ids = [], []
        for doc_id, info in self.documents.items():
            corpus.append(" ".join(ch["chunk"] for ch in info.get("chunks", [])))
            self.tfidf_doc_ids.append(doc_id)
        if corpus:
            vectorizer = TfidfVectorizer()
            self.tfidf_index = vectorizer.fit_transform(corpus)
            self.tfidf_vectorizer = vectorizer
        else:
            self.tfidf_index = None
            self.tfidf_vectorizer = None

    def get_tfidf_index(self):
        return self.tfidf_vectorizer, self.tfidf_index, self.tfidf_doc_ids

    def get_tfidf_scorer(self):
        """HBM-resident scorer for the current (vectorizer, matrix); None without a TF-IDF index."""
        if self.tfidf_vectorizer is None or self.tfidf_index is None:
            return None
        key = (id(self.tfidf_vectorizer), id(self.tfidf_index))
        if self._tfidf_scorer is None or self._tfidf_scorer_key != key:
            ctx = getattr(self, "_shard_ctx", None)
            if ctx is not None:     # this rank's block of the document rows (rag_system_amd.sharded)
                from .sharded import ShardedTfidfScorer, shard_bounds
                n_docs = self.tfidf_index.shape[0]
                lo, cnt = shard_bounds(n_docs, ctx.world)[ctx.rank]
                local = TfidfScorer(self.tfidf_vectorizer, self.tfidf_index, shard=(lo, cnt)) if cnt else None
                self._tfidf_scorer = ShardedTfidfScorer(local, lo, n_docs, ctx, self.tfidf_vectorizer)
            else:
                self._tfidf_scorer = TfidfScorer(self.tfidf_vectorizer, self.tfidf_index)
            self._tfidf_scorer_key = key
        return self._tfidf_scorer

    # ---- row sharding over the GPUs of one node (one process per GPU) ---------------------------

    def shard(self, ctx=None, dense_rows=None):
        """Turn this store -- complete host metadata, as every rank loads or builds it -- into ONE RANK'S PART of a
        row-sharded store: the dense index keeps the contiguous row block of this rank (global ids = block start
        + local row), the BM25 and TF-IDF indices the postings of this rank's block of chunk / document rows, with
        the statistics (idf, avgdl) of the whole corpus.  search / search_bm25 / the TF-IDF scorer then exchange
        their per-rank top lists with one all-gather and merge them: every rank returns what the unsharded store
        returns (tests/test_sharded_gloo.py).  SPMD: every rank calls the same methods with the same arguments.
        ctx: rag_system_amd.sharded.ShardContext (default: torch.distributed's default group).
        A sharded store is searched and appended to; save() needs the whole matrix and is refused."""
        from .sharded import ShardContext, ShardedBM25, ShardedIndexFlat, shard_bounds
        ctx = ctx or ShardContext()
        if getattr(self.index, "sharded", False):
            raise ValueError("this store is sharded already")
        n = self.index.ntotal if dense_rows is None else dense_rows[1]
        lo, cnt = shard_bounds(n, ctx.world)[ctx.rank]
        if dense_rows is None:      # the index holds every row: keep this rank's block
            local = faiss.IndexFlat(self.embedding_dim, self.index.metric_type)
            if cnt:
                local.add(self.index.reconstruct_n(lo, cnt))
        else:                       # load(shard=...): only the block was read from index.faiss
            local = self.index
        if self._dense_prefilter:
            local.set_prefilter(True)
        self.index = ShardedIndexFlat(local, lo, n, ctx)
        self._shard_ctx = ctx
        if self.bm25 is not None:
            nb = len(self.bm25_corpus)
            blo, bcnt = shard_bounds(nb, ctx.world)[ctx.rank]
            local_bm = BM25Okapi(self.bm25_corpus, k1=self.bm25.k1, b=self.bm25.b, shard=(blo, bcnt))
            self.bm25 = ShardedBM25(local_bm, blo, nb, ctx)
        self._tfidf_scorer = None       # rebuilt sharded by get_tfidf_scorer
        self._tfidf_scorer_key = None
        return self

    def hybrid_doc_tables(self, scorer):
        """Side tables of rag_hybrid_rrf_mmr's MMR stage for the current indices, uploaded once (SURVEY 8(f) f4
        on the device): per TF-IDF row, the row of the dense index that IS the document's embedding -- a
        document with exactly one stored chunk, whose text mmr_selection would re-encode (search_engine.py:415,
        :418) -- else -1, and the UTF-8 length of the document text (:437, :458).  False when the dense index
        and chunks_data are out of step or the scorer has no such call."""
        if scorer is None or not hasattr(scorer, "set_doc_tables") or self.index.ntotal != len(self.chunks_data):
            return False
        key = (id(scorer), id(self.tfidf_doc_ids), len(self.tfidf_doc_ids), len(self.chunks_data), id(self.index))
        if getattr(self, "_hybrid_tables_key", None) == key:
            return self._hybrid_tables_ok
        first_row, n_chunks = {}, {}
        for i, cd in enumerate(self.chunks_data):
            d = cd["doc_id"]
            n_chunks[d] = n_chunks.get(d, 0) + 1
            first_row.setdefault(d, i)
        flat_row = np.full(len(self.tfidf_doc_ids), -1, dtype=np.int64)
        text_bytes = np.zeros(len(self.tfidf_doc_ids), dtype=np.int64)
        for r, d in enumerate(self.tfidf_doc_ids):
            chunks = self.documents.get(d, {}).get("chunks", [])
            text_bytes[r] = len(" ".join(ch["chunk"] for ch in chunks).encode("utf-8"))
            # the document's text must be the text of the one chunk the index holds for it
            if n_chunks.get(d) == 1 and len(chunks) == 1 and chunks[0]["chunk"] == self.chunks_data[first_row[d]]["chunk"]:
                flat_row[r] = first_row[d]
        self._hybrid_tables_ok = bool(scorer.set_doc_tables(flat_row, text_bytes))
        self._hybrid_tables_key = key
        return self._hybrid_tables_ok

    # ---- maintenance ------------------------------------------------------------------

    def rebuild_all_indices(self, embedding_manager):
        self.rebuild_faiss_index(embedding_manager)
        if "bm25" in self.indices_present:
            self.rebuild_bm25_index()
        if "tfidf" in self.indices_present:
            self.rebuild_tfidf_index()

    def remove_document(self, doc_id, embedding_manager):
        if doc_id not in self.documents:
            return False
        del self.documents[doc_id]
        kept = [(cid, cd) for cid, cd in zip(self.doc_ids, self.chunks_data) if cid != doc_id]
        self.doc_ids = [cid for cid, _ in kept]
        self.chunks_data = [cd for _, cd in kept]
        for sig in [s for s, d in self.signatures.items() if d == doc_id]:
            del self.signatures[sig]
        self.rebuild_all_indices(embedding_manager)
        return True

    # ---- persistence (same four files, same keys) ---------------------------------------

    def save(self, storage_dir, embedding_manager, reencode=True):
        """reencode=True is the reference (vector_storage.py:227 -> rebuild_faiss_index re-encodes EVERY chunk
        with model.encode, :148-152).  reencode=False writes the vectors the index already
        holds in HBM -- the ones add_vector received -- and only rebuilds the BM25 / TF-IDF indices; the files
        have the same layout, the dense rows differ from a re-encoded save only where the reference's own
        quirk makes them differ (normalised adds vs raw model outputs)."""
        if getattr(self.index, "sharded", False):
            raise ValueError("a row-sharded store cannot be saved (each rank holds a block of the rows): save from an "
                             "unsharded store")
        if reencode or self.index.ntotal != len(self.chunks_data):
            self.rebuild_all_indices(embedding_manager)
        else:
            if "bm25" in self.indices_present:
                self.rebuild_bm25_index()
            if "tfidf" in self.indices_present:
                self.rebuild_tfidf_index()
        os.makedirs(storage_dir, exist_ok=True)
        paths = {name: os.path.join(storage_dir, fn) for name, fn in _FILES.items()}
        faiss.write_index(self.index, paths["faiss"])
        if "bm25" in self.indices_present and self.bm25 is not None:
            with open(paths["bm25"], "wb") as f:
                pickle.dump({"bm25_corpus": self.bm25_corpus, "bm25_doc_ids": self.bm25_doc_ids,
                             "bm25_indexed_count": self.bm25_indexed_count}, f)
        elif os.path.exists(paths["bm25"]):
            os.remove(paths["bm25"])
        if "tfidf" in self.indices_present and self.tfidf_index is not None:
            with open(paths["tfidf"], "wb") as f:
                pickle.dump({"tfidf_index": self.tfidf_index, "tfidf_vectorizer": self.tfidf_vectorizer,
                             "tfidf_doc_ids": self.tfidf_doc_ids}, f)
        elif os.path.exists(paths["tfidf"]):
            os.remove(paths["tfidf"])
        with open(paths["meta"], "wb") as f:
            pickle.dump({"embedding_dim": self.embedding_dim, "embedder": self.embedder, "doc_ids": self.doc_ids,
                         "chunks_data": self.chunks_data, "documents": self.documents,
                         "signatures": self.signatures, "indices_present": self.indices_present,
                         "chunk_size": self.chunk_size, "overlap": self.overlap,
                         # not in the reference (its load() ignores unknown keys): which tokenizer split the
                         # BM25 corpus, so that a load with another one can say so
                         "bm25_tokenizer": _text.tokenizer_name()}, f)
        return paths["faiss"], paths["bm25"], paths["tfidf"], paths["meta"]

    @classmethod
    def load(cls, storage_dir, bm25_params="reference", shard=None):
        """Stores are the caller's own files (written by save() here or by the reference).

        bm25_params: the reference scores a FRESH store with Config's k1 = 1.0, b = 0.65
        (vector_storage.py:170) but a LOADED one with rank_bm25's defaults k1 = 1.5, b = 0.75 (:305 passes
        no parameters), so the same query ranks differently before and after a save / load cycle.
        "reference" keeps that behaviour; "config" rebuilds with Config.BM25_PARAM_K1 / _B like a fresh
        store; a (k1, b) tuple sets them explicitly."""
        if bm25_params == "reference":
            bm25_kw = {}
        elif bm25_params == "config":
            bm25_kw = {"k1": Config.BM25_PARAM_K1, "b": Config.BM25_PARAM_B}
        elif isinstance(bm25_params, (tuple, list)) and len(bm25_params) == 2:
            bm25_kw = {"k1": float(bm25_params[0]), "b": float(bm25_params[1])}
        else:
            raise ValueError("bm25_params must be 'reference', 'config' or a (k1, b) pair, got %r" % (bm25_params,))
        paths = {name: os.path.join(storage_dir, fn) for name, fn in _FILES.items()}
        with open(paths["meta"], "rb") as f:
            meta = safe_pickle.load_plain(f, "metadata.pkl")     # restricted unpickling: safe_pickle.py
        indices_present = meta.get("indices_present", ["faiss"])
        storage = cls(embedding_dim=meta["embedding_dim"], embedder=meta.get("embedder", "all-MiniLM-L6-v2"),
                      indices_present=indices_present, chunk_size=meta.get("chunk_size", 100),
                      overlap=meta.get("overlap", 20))
        storage.doc_ids = meta.get("doc_ids", [])
        storage.chunks_data = meta.get("chunks_data", [])
        storage.documents = meta.get("documents", {})
        storage.signatures = meta.get("signatures", {})
        ctx = None
        if shard is not None and shard is not False:
            # shard=True / a ShardContext: this rank reads only ITS block of index.faiss (sharded below)
            from .sharded import ShardContext, shard_bounds
            ctx = shard if isinstance(shard, ShardContext) else ShardContext()
            n_file = faiss.read_index(paths["faiss"], 0, 0).file_ntotal
            lo, cnt = shard_bounds(n_file, ctx.world)[ctx.rank]
            storage.index = faiss.read_index(paths["faiss"], lo, cnt)
        else:
            storage.index = faiss.read_index(paths["faiss"])
        if "bm25" in indices_present and os.path.exists(paths["bm25"]):
            with open(paths["bm25"], "rb") as f:
                data = safe_pickle.load_plain(f, "index_bm25.pkl")
            # the corpus was tokenised when it was indexed; queries are tokenised now: same tokenizer?
            _text.check_store_tokenizer(meta.get("bm25_tokenizer"))
            storage.bm25_corpus = data["bm25_corpus"]
            storage.bm25_doc_ids = data["bm25_doc_ids"]
            storage.bm25_indexed_count = data.get("bm25_indexed_count", len(storage.chunks_data))
            storage.bm25 = BM25Okapi(storage.bm25_corpus, **bm25_kw)   # no parameters = library defaults (:305)
        if "tfidf" in indices_present and os.path.exists(paths["tfidf"]):
            with open(paths["tfidf"], "rb") as f:
                data = safe_pickle.load_tfidf(f)
            storage.tfidf_index = data["tfidf_index"]
            storage.tfidf_vectorizer = data["tfidf_vectorizer"]
            storage.tfidf_doc_ids = data["tfidf_doc_ids"]
        if ctx is not None:
            storage.shard(ctx, dense_rows=(lo, n_file))
        return storage


class VectorStorageManager:
    def __init__(self):
        self.storages = {}

    def create_storage(self, name, embedding_dim, embedder="all-MiniLM-L6-v2"):
        if name in self.storages:
            raise ValueError("Esiste già uno storage con nome '%s'." % name)
        self.storages[name] = VectorStorage(embedding_dim, embedder=embedder,
                                            indices_present=["faiss", "bm25", "tfidf"])
        return self.storages[name]

    def get_storage(self, name):
        return self.storages.get(name)

    def list_storages(self):
        return list(self.storages.keys())

    def delete_storage(self, name, base_dir):
        self.storages.pop(name, None)
        storage_dir = os.path.join(base_dir, name)
        if os.path.exists(storage_dir):
            shutil.rmtree(storage_dir)
        return True

    def save_storage(self, name, base_dir, embedding_manager, reencode=True):
        storage = self.get_storage(name)
        if not storage:
            raise ValueError("Storage '%s' non trovato." % name)
        storage.set_chunk_params(embedding_manager.chunk_size, embedding_manager.overlap)
        storage.save(os.path.join(base_dir, name), embedding_manager, reencode=reencode)

    def load_storage(self, name, base_dir, bm25_params="reference", shard=None):
        """shard (not in the reference): None = Config.ENGINE_SHARD ("auto": shard over torch.distributed's ranks
        when a process group with more than one rank is initialised), True / a ShardContext = shard, False = not."""
        if shard is None:
            shard = Config.ENGINE_SHARD
        if shard == "auto":
            shard = False
            try:
                import sys
                dist = sys.modules.get("torch.distributed")     # never imports torch by itself
                shard = bool(dist is not None and dist.is_initialized() and dist.get_world_size() > 1)
            except Exception:
                shard = False
        storage = VectorStorage.load(os.path.join(base_dir, name), bm25_params=bm25_params, shard=shard or None)
        self.storages[name] = storage
        return storage

    def list_persistent_storages(self, base_dir):
        found = []
        if not os.path.exists(base_dir):
            return found
        for folder in os.listdir(base_dir):
            meta_path = os.path.join(base_dir, folder, _FILES["meta"])
            if not os.path.isfile(meta_path):
                continue
            try:
                with open(meta_path, "rb") as f:
                    meta = safe_pickle.load_plain(f, "metadata.pkl")
                found.append({"name": folder, "embedding_dim": meta.get("embedding_dim", 384),
                              "doc_ids": meta.get("doc_ids", []), "document_count": len(meta.get("documents", {}))})
            except Exception:
                pass
        return found
