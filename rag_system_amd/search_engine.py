"""SearchEngine with the reference's strategies, scored by the MI355X engine.

Mirrors rag_system/search_engine.py: detect_knee :37, SearchEngine :51 with
search_with_strategy :76, search_faiss :138, search_bm25 :174, search_hybrid :195,
search_multi_representation :272, search_rrf :320, reciprocal_rank_fusion :352,
rerank_with_cross_encoder :362, mmr_selection :396.  Same arguments, same result dictionaries
({doc_id, score, chunk, source} / {doc_id, score, text[, source]}, optional ce_score), same
ordering rules (Python's stable descending sort => ties keep first-seen order).

Where the reference loops over ALL documents in Python for the TF-IDF signal (dict over every
doc :286, full sort :332-336), this engine keeps the score vector in HBM and asks it only for
what the fusion can possibly use -- with identical results:
  * a document outside the dense/BM25 candidate set A gets, in `rrf`, exactly 1/(60+rank) and,
    in `multi`, exactly w_tfidf * score/max: both strictly ordered by its TF-IDF rank, so at most
    the first top_k + |A| documents of the TF-IDF ranking can reach the final top_k;
  * members of A need their own TF-IDF rank (rag_csc_ranks) or score (rag_csc_gather).
Arithmetic on the (tiny) candidate set is done in Python floats in the reference's order.
"""
from typing import Any, Dict, List

import numpy as np

from . import candidates


def detect_knee(scores: List[float]) -> float:
    """Score right after the largest drop between consecutive (descending) scores."""
    if len(scores) < 2:
        return scores[0] if scores else 0.0
    drops = [scores[i] - scores[i + 1] for i in range(len(scores) - 1)]
    return scores[drops.index(max(drops)) + 1]


def _rank_key(r):
    return r.get("ce_score", r.get("score", 0.0))


class _Begun:
    """A begun dense search: the store's token and the query embedding it was begun with."""
    __slots__ = ("tok", "q_vec")

    def __init__(self, tok, q_vec):
        self.tok, self.q_vec = tok, q_vec


class SearchEngine:
    def __init__(self, vector_storage, embedding_manager, cross_encoder=None, weight_dense=0.4, weight_bm25=0.3,
                 weight_tfidf=0.3, verbose=False):
        self.vector_storage = vector_storage
        self.embedding_manager = embedding_manager
        self.cross_encoder = cross_encoder
        self.weight_dense = weight_dense
        self.weight_bm25 = weight_bm25
        self.weight_tfidf = weight_tfidf
        self.verbose = verbose

    def _debug(self, msg):
        if self.verbose:
            print("[DEBUG] " + msg)

    # ------------------------------------------------------------------ dispatch

    def search_with_strategy(self, query: str, strategy="faiss", top_k=10, retrieval_mode="chunk", threshold=0.0,
                             use_knee_detection: bool = False, use_mmr: bool = False, mmr_lambda: float = 0.5,
                             max_context_bytes: int = None, mmr_reuse_index_vectors: bool = False) -> List[Dict[str, Any]]:
        strategy = strategy.lower().strip()
        retrieval_mode = retrieval_mode.lower().strip()
        self._debug("search_with_strategy strategy=%s top_k=%s mode=%s threshold=%s"
                    % (strategy, top_k, retrieval_mode, threshold))
        if strategy == "faiss":
            results = self.search_faiss(query, top_k, retrieval_mode)
        elif strategy == "bm25":
            results = self.search_bm25(query, top_k, retrieval_mode)
        elif strategy in ("ibrido", "hybrid"):
            results = self.search_hybrid(query, top_k, retrieval_mode)
        elif strategy == "multi":
            results = self.search_multi_representation(query, top_k, "document")
        elif strategy == "rrf":
            # everything behind the three rankings -- fusion, sort, cut, knee, threshold and (with
            # mmr_reuse_index_vectors) MMR -- in one engine call when it is served (rag_hybrid_rrf_mmr)
            post = None
            if not self.cross_encoder:
                post = {"threshold": threshold, "knee": bool(use_knee_detection),
                        "mmr": bool(use_mmr and mmr_reuse_index_vectors), "lambda": mmr_lambda,
                        "max_bytes": max_context_bytes}
            results, final = self._search_rrf(query, top_k, post)
            if final is not None:
                kept, order = final
                if not use_mmr or not kept:
                    return kept
                if order is not None:
                    return [kept[i] for i in order]
                return self.mmr_selection(query, kept, self.embedding_manager, lambda_param=mmr_lambda, top_k=top_k,
                                          max_context_bytes=max_context_bytes, reuse_index_vectors=mmr_reuse_index_vectors)
        else:
            raise ValueError(f"Unknown retrieval strategy: {strategy}")
        results.sort(key=_rank_key, reverse=True)
        if use_knee_detection and results:
            cutoff = detect_knee([_rank_key(r) for r in results])
            results = [r for r in results if _rank_key(r) > cutoff]
        kept = [r for r in results if _rank_key(r) >= threshold]
        if use_mmr and kept:
            kept = self.mmr_selection(query, kept, self.embedding_manager, lambda_param=mmr_lambda, top_k=top_k,
                                      max_context_bytes=max_context_bytes, reuse_index_vectors=mmr_reuse_index_vectors)
        return kept

    # ------------------------------------------------------------------ helpers

    def _doc_text(self, doc_id):
        info = self.vector_storage.documents.get(doc_id, {})
        return " ".join(ch["chunk"] for ch in info.get("chunks", []))

    @staticmethod
    def _group_dense_by_doc(chunk_hits):
        """doc_id -> {'score': max chunk score, 'chunks': chunk texts in hit order}; keys in first-hit order."""
        groups = {}
        for r in chunk_hits:
            g = groups.get(r["doc_id"])
            if g is None:
                groups[r["doc_id"]] = {"score": r["score"], "chunks": [r["chunk"]]}
            else:
                g["score"] = max(g["score"], r["score"])
                g["chunks"].append(r["chunk"])
        return groups

    def _dense_chunks_begin(self, query, k):
        """The dense pass in two halves: the reference computes the dense ranking and then the BM25
        ranking of the same query (search_engine.py:266-267, :230-231, :170-173); they do not depend
        on each other, so the flat search is enqueued here (rag_flat_search_begin) and the sparse
        scoring runs beside it.  Stores without the split (test backends) search at once."""
        q_vec = self.embedding_manager.embed_query(query)
        vs = self.vector_storage
        if hasattr(vs, "search_begin"):
            return _Begun(vs.search_begin(q_vec, top_k=k), q_vec)
        return _Begun(("done", vs.search(q_vec, top_k=k)), q_vec)

    def _dense_chunks_end(self, token):
        vs = self.vector_storage
        hits = vs.search_end(token.tok) if hasattr(vs, "search_end") else token.tok[1]
        for r in hits:
            r["score"] = float(r["score"])
        return hits

    def _dense_pairs_end(self, token):
        """[(doc_id, score)] of a begun dense search for stores that offer it (search_end(ids_only=True)), else
        None (nothing was collected yet)."""
        vs = self.vector_storage
        if not hasattr(vs, "hybrid_doc_tables"):     # not rag_system_amd's VectorStorage (test fakes): the dict route
            return None
        return vs.search_end(token.tok, ids_only=True)

    def _dense_abort(self, token):
        """A stage between the two halves raised: collect the begun dense search (its result is dropped) so
        that the index accepts the next one."""
        try:
            if token is not None:
                self._dense_chunks_end(token)
        except Exception:
            pass

    def _dense_chunks(self, query, k):
        return self._dense_chunks_end(self._dense_chunks_begin(query, k))

    # ------------------------------------------------------------------ single-signal strategies

    def search_faiss(self, query: str, top_k=10, retrieval_mode="chunk") -> List[Dict[str, Any]]:
        return self._search_faiss_end(self._search_faiss_begin(query, top_k, retrieval_mode))

    def _search_faiss_begin(self, query, top_k, retrieval_mode):
        if "faiss" not in self.vector_storage.indices_present:
            return None
        k = top_k * 3 if retrieval_mode == "document" else top_k
        return (self._dense_chunks_begin(query, k), query, top_k, retrieval_mode)

    def _search_faiss_end(self, pending, want_text=True):
        """want_text=False (fusion strategies without a cross-encoder): the joined chunk texts of the document
        ranking are never looked at again (the fused entries get the full document text, :343-345), so they
        are not built."""
        if pending is None:
            return []
        token, query, top_k, retrieval_mode = pending
        if retrieval_mode == "document" and not want_text and not self.cross_encoder:
            pairs = self._dense_pairs_end(token)
            if pairs is not None:       # per-document maximum in first-hit order (:148-155), no dictionaries per hit
                best = {}
                for d, sc in pairs:
                    if d not in best or sc > best[d]:
                        best[d] = sc
                results = [{"doc_id": d, "score": sc} for d, sc in best.items()]
                results.sort(key=lambda x: x["score"], reverse=True)
                return results[:top_k]
        hits = self._dense_chunks_end(token)
        if retrieval_mode == "document":
            groups = self._group_dense_by_doc(hits)
            if want_text or self.cross_encoder:
                results = [{"doc_id": d, "score": g["score"], "text": " ".join(g["chunks"])} for d, g in groups.items()]
            else:
                results = [{"doc_id": d, "score": g["score"]} for d, g in groups.items()]
        else:
            results = hits
        if self.cross_encoder:
            results = self.rerank_with_cross_encoder(query, results, normalize_ce=True)
        results.sort(key=lambda x: x["score"], reverse=True)
        return results[:top_k]

    def search_faiss_batch(self, queries, top_k=10, retrieval_mode="chunk") -> List[List[Dict[str, Any]]]:
        """search_faiss for a list of queries with ONE index.search(nq x d) call -- the batched
        front end the reference lacks (search_engine.py:141 is strictly one query).
        Element i equals search_faiss(queries[i], top_k, retrieval_mode) up to near-ties between
        the batch and single-query kernels (both within the 1e-5 dense tolerance)."""
        queries = list(queries)
        if "faiss" not in self.vector_storage.indices_present or not queries:
            return [[] for _ in queries]
        k = top_k * 3 if retrieval_mode == "document" else top_k
        q_vecs = self._embed_queries(queries)
        out = []
        for query, hits in zip(queries, self.vector_storage.search_batch(q_vecs, top_k=k)):
            for r in hits:
                r["score"] = float(r["score"])
            if retrieval_mode == "document":
                groups = self._group_dense_by_doc(hits)
                results = [{"doc_id": d, "score": g["score"], "text": " ".join(g["chunks"])} for d, g in groups.items()]
            else:
                results = hits
            if self.cross_encoder:
                results = self.rerank_with_cross_encoder(query, results, normalize_ce=True)
            results.sort(key=lambda x: x["score"], reverse=True)
            out.append(results[:top_k])
        return out

    def _embed_queries(self, queries):
        """(nq, d) float32 query embeddings with ONE model.encode call for the whole batch, normalised row by row
        as embed_query does (embedding_manager.py:88-98: float32, divided by the norm, a norm below 1e-9 counts as
        1).  Managers that offer embed_queries are asked directly; managers without a batch-capable model fall
        back to embed_query per query."""
        em = self.embedding_manager
        batch = getattr(em, "embed_queries", None)
        if batch is not None:
            return np.ascontiguousarray(batch(queries), dtype="float32").reshape(len(queries), -1)
        model = getattr(em, "model", None)
        if model is not None and hasattr(model, "encode"):
            try:
                vecs = np.array(model.encode(list(queries)), dtype="float32")
            except Exception as e:       # a model that takes one text at a time
                self._debug("batched encode failed (%s): one embed_query per query" % e)
                vecs = None
            if vecs is not None and vecs.ndim == 2 and vecs.shape[0] == len(queries):
                for i in range(len(vecs)):
                    norm = np.linalg.norm(vecs[i])
                    if norm < 1e-9:
                        norm = 1.0
                    vecs[i] /= norm
                return vecs
        return np.stack([np.asarray(em.embed_query(q), dtype="float32").reshape(-1) for q in queries])

    def search_bm25(self, query: str, top_k=10, retrieval_mode="document", _want_text=True) -> List[Dict[str, Any]]:
        if "bm25" not in self.vector_storage.indices_present:
            return []
        hits = self.vector_storage.search_bm25(query, top_k=top_k)
        if retrieval_mode == "document":
            # one entry per chunk hit, doc ids may repeat
            if not _want_text and not self.cross_encoder:     # fusion strategies: only ids and scores are used
                return [{"doc_id": r["doc_id"], "score": r["score"]} for r in hits]
            docs = [{"doc_id": r["doc_id"], "score": r["score"], "text": self._doc_text(r["doc_id"])} for r in hits]
            if self.cross_encoder:
                docs = self.rerank_with_cross_encoder(query, docs, normalize_ce=False)
            return docs
        if self.cross_encoder:
            hits = self.rerank_with_cross_encoder(query, hits, normalize_ce=True)
        return hits

    # ------------------------------------------------------------------ hybrid (dense + bm25, raw scores)

    def search_hybrid(self, query: str, top_k=10, retrieval_mode="chunk") -> List[Dict[str, Any]]:
        pending = None
        if "faiss" in self.vector_storage.indices_present:
            pending = self._dense_chunks_begin(query, top_k * 3 if retrieval_mode == "document" else top_k)
        sparse = []
        if "bm25" in self.vector_storage.indices_present:
            try:
                sparse = self.search_bm25(query, top_k=top_k, retrieval_mode=retrieval_mode)   # beside the dense pass
            except Exception:
                self._dense_abort(pending)
                raise
        dense = self._dense_chunks_end(pending) if pending is not None else []
        fused = {}
        if retrieval_mode == "document":
            groups = self._group_dense_by_doc(dense)
            for doc_id, g in groups.items():
                fused[doc_id] = self.weight_dense * g["score"]
            for r in sparse:
                fused[r["doc_id"]] = fused.get(r["doc_id"], 0) + self.weight_bm25 * r["score"]
            results = [{"doc_id": d, "score": s,
                        "text": " ".join(groups[d]["chunks"]) if d in groups else self._doc_text(d),
                        "source": "hybrid"} for d, s in fused.items()]
        else:
            best = {}   # doc_id -> (chunk text, score) of its best dense chunk (first wins ties)
            for r in dense:
                if r["doc_id"] not in best or r["score"] > best[r["doc_id"]][1]:
                    best[r["doc_id"]] = (r["chunk"], r["score"])
            for doc_id, (_txt, sc) in best.items():
                fused[doc_id] = self.weight_dense * sc
            for r in sparse:
                if r["doc_id"] in best:   # bm25 only boosts documents the dense search found
                    fused[r["doc_id"]] = fused.get(r["doc_id"], 0) + self.weight_bm25 * r["score"]
            results = [{"doc_id": d, "score": s, "chunk": best[d][0] if d in best else "", "source": "hybrid"}
                       for d, s in fused.items()]
        results.sort(key=lambda x: x["score"], reverse=True)
        results = results[:top_k]
        if self.cross_encoder:
            results = self.rerank_with_cross_encoder(query, results, normalize_ce=True)
        return results

    # ------------------------------------------------------------------ TF-IDF access

    def _tfidf_begin(self, query, k_bound):
        """Enqueue the TF-IDF scoring of `query` (rag_tfidf_query_begin) so that it runs beside the dense pass
        and the BM25 scoring: the reference scores TF-IDF after both (search_engine.py:330-331), but the scores
        depend on the query alone.  Returns a token for _tfidf_for_candidates, or None (nothing begun)."""
        try:
            scorer = self.vector_storage.get_tfidf_scorer()
            begin = getattr(scorer, "query_begin", None)
            if scorer is None or begin is None:
                return None
            tok = begin(query, k_bound)
            return None if tok is None else (scorer, tok)
        except Exception as e:
            self._debug("TF-IDF begin failed: %s" % e)
            return None

    def _tfidf_abort(self, begun):
        if begun is not None:
            try:
                begun[0].query_abort(begun[1])
            except Exception:
                pass

    def _tfidf_for_candidates(self, query, cand_doc_ids, n_extra, begun=None):
        """Score the query against the HBM TF-IDF index once and return
        (top_ids, top_scores): the first `n_extra` documents of the stable descending ranking,
        (cand_rank, cand_score): TF-IDF rank / score of every candidate doc id that is indexed.
        None if the store has no TF-IDF index."""
        try:
            scorer = self.vector_storage.get_tfidf_scorer()
            doc_ids = self.vector_storage.tfidf_doc_ids
        except Exception as e:
            self._debug("get_tfidf_scorer failed: %s" % e)
            scorer, doc_ids = None, []
        if scorer is None:
            self._tfidf_abort(begun)
            return None
        pos = self._tfidf_rows(None) or {}
        rows = [pos[d] for d in cand_doc_ids if d in pos]
        k = min(n_extra, len(doc_ids))
        if begun is not None and begun[0] is scorer and k <= begun[1][1] and rows:
            top_idx, top_sc, ranks, scores = scorer.query_end(begun[1], k, rows)
        else:
            self._tfidf_abort(begun)
            top_idx, top_sc, ranks, scores = scorer.query(query, k, rows)
        cand = {doc_ids[r]: (int(rk), float(sc)) for r, rk, sc in zip(rows, ranks, scores)}
        return [doc_ids[int(i)] for i in top_idx], [float(s) for s in top_sc], cand

    # ------------------------------------------------------------------ multi (dense + bm25 + tfidf, max-normalised)

    def search_multi_representation(self, query: str, top_k=10, retrieval_mode="document") -> List[Dict[str, Any]]:
        pending = self._search_faiss_begin(query, top_k * 3, "document")
        tf_begun = self._tfidf_begin(query, top_k * 7)     # beside the dense pass, like the BM25 scoring below
        try:
            bm25_hits = self.search_bm25(query, top_k=top_k * 3, retrieval_mode="document", _want_text=False)
        except Exception:
            self._tfidf_abort(tf_begun)
            self._dense_abort(pending[0] if pending else None)
            raise
        try:
            dense = {d["doc_id"]: d["score"] for d in self._search_faiss_end(pending, want_text=False)}
        except Exception:
            self._tfidf_abort(tf_begun)
            raise
        bm25 = {d["doc_id"]: d["score"] for d in bm25_hits}
        cand_ids = list(dict.fromkeys(list(dense) + list(bm25)))
        tf = self._tfidf_for_candidates(query, cand_ids, top_k + len(cand_ids), tf_begun)
        tfidf = {}
        tfidf_max = None
        if tf is not None:
            top_ids, top_sc, cand = tf
            tfidf_max = top_sc[0] if top_sc else None          # max over ALL documents
            for d in cand_ids:
                if d in cand:
                    tfidf[d] = cand[d][1]
            for d, s in zip(top_ids, top_sc):
                tfidf.setdefault(d, s)

        def normalised(scores, mx=None):
            if not scores:
                return {}
            mx = max(scores.values()) if mx is None else mx
            return {k: v / mx for k, v in scores.items()} if mx > 0 else scores

        dn, bn, tn = normalised(dense), normalised(bm25), normalised(tfidf, tfidf_max)
        results = []
        for doc_id in dict.fromkeys(list(dn) + list(bn) + list(tn)):
            fused = (self.weight_dense * dn.get(doc_id, 0) + self.weight_bm25 * bn.get(doc_id, 0)
                     + self.weight_tfidf * tn.get(doc_id, 0))
            results.append({"doc_id": doc_id, "score": fused})
        results.sort(key=lambda x: x["score"], reverse=True)
        results = results[:top_k]
        for r in results:                   # the text only for what survives the cut (stable sort: same order)
            r["text"] = self._doc_text(r["doc_id"])
        if self.cross_encoder:
            results = self.rerank_with_cross_encoder(query, results, normalize_ce=True)
        return results

    # ------------------------------------------------------------------ rrf

    def search_rrf(self, query: str, top_k=10, retrieval_mode="document") -> List[Dict[str, Any]]:
        return self._search_rrf(query, top_k, None)[0]

    def _tfidf_rows(self, doc_ids):
        """TF-IDF row of every doc id (built once per store), or None without a TF-IDF index."""
        vs = self.vector_storage
        doc_list = getattr(vs, "tfidf_doc_ids", None)
        if not doc_list:
            return None
        pos = getattr(vs, "_tfidf_pos", None)
        if pos is None or pos[0] is not doc_list or len(pos[1]) != len(doc_list):
            pos = (doc_list, {d: i for i, d in enumerate(doc_list)})
            vs._tfidf_pos = pos
        return pos[1]

    def _search_rrf(self, query, top_k, post):
        """search_rrf (:320-350).  post = None: plain.  post = {threshold, knee, mmr, lambda, max_bytes} (from
        search_with_strategy, no cross-encoder): what follows the fusion there (:121-133) may run in the same
        engine call; then the second return value is (kept results, MMR order or None), else it is None and
        the first is search_rrf's result."""
        tables_ok = False
        if post is not None and post["mmr"]:
            # the MMR side tables are uploaded once per store, and before a TF-IDF query is begun
            try:
                tables = getattr(self.vector_storage, "hybrid_doc_tables", None)
                tables_ok = bool(tables is not None and tables(self.vector_storage.get_tfidf_scorer()))
            except Exception as e:
                self._debug("hybrid_doc_tables failed: %s" % e)
        pending = self._search_faiss_begin(query, top_k * 3, "document")
        tf_begun = self._tfidf_begin(query, top_k * 7)     # beside the dense pass, like the BM25 scoring below
        try:
            bm25_ranking = self.search_bm25(query, top_k=top_k * 3, retrieval_mode="document", _want_text=False)
        except Exception:
            self._tfidf_abort(tf_begun)
            self._dense_abort(pending[0] if pending else None)
            raise
        try:
            dense_ranking = self._search_faiss_end(pending, want_text=False)
        except Exception:
            self._tfidf_abort(tf_begun)
            raise
        ordered = [[r["doc_id"] for r in sorted(ranking, key=lambda x: x["score"], reverse=True)]
                   for ranking in (dense_ranking, bm25_ranking)]
        # ---- one engine call for everything behind the rankings (rag_hybrid_rrf_mmr), when it is served
        if tf_begun is not None:
            try:
                done = self._rrf_tail(query, top_k, post, pending, tf_begun, ordered, tables_ok)
            except Exception:
                self._tfidf_abort(tf_begun)
                raise
            if done is not None:
                return done
        # ---- the three-call route: rag_rrf, TF-IDF rank query, fusion on the host
        try:
            fused = candidates.reciprocal_rank_fusion(ordered, k=60)
        except Exception:
            self._tfidf_abort(tf_begun)
            raise
        tf = self._tfidf_for_candidates(query, list(fused), top_k + len(fused), tf_begun)
        if tf is not None:
            top_ids, _top_sc, cand = tf
            for doc_id in fused:                       # candidates: their own rank among ALL documents
                if doc_id in cand:
                    fused[doc_id] = fused[doc_id] + 1.0 / (60 + cand[doc_id][0])
            for rank, doc_id in enumerate(top_ids):    # the head of the TF-IDF ranking, in ranking order
                if doc_id not in fused:
                    fused[doc_id] = 0 + 1.0 / (60 + rank)
        combined = sorted(fused.items(), key=lambda kv: kv[1], reverse=True)[:top_k]     # stable, like :346-347
        combined = [{"doc_id": d, "score": s, "text": self._doc_text(d)} for d, s in combined]
        if self.cross_encoder:
            combined = self.rerank_with_cross_encoder(query, combined, normalize_ce=True)
        return combined, None

    def _rrf_tail(self, query, top_k, post, pending, tf_begun, ordered, tables_ok=False):
        """The fused tail.  None = not served (the TF-IDF query is still begun); else _search_rrf's result."""
        scorer, tok = tf_begun
        if not hasattr(scorer, "hybrid_tail") or top_k > 64 or len(ordered[0]) > 128 or len(ordered[1]) > 128:
            return None
        pos = self._tfidf_rows(None)
        if pos is None:
            return None
        doc_list = self.vector_storage.tfidf_doc_ids
        extra = {}                                      # documents without a TF-IDF row: codes -1, -2, ...
        rows = [[pos[d] if d in pos else -1 - extra.setdefault(d, len(extra)) for d in ranking] for ranking in ordered]
        n_cand = len(set(rows[0]) | set(rows[1]))
        if top_k + n_cand > tok[1] and tok[1] < len(doc_list):
            return None                                 # the begun selection is too short for this candidate set
        in_kernel = post is not None
        flat = q_vec = None
        if in_kernel and post["mmr"] and tables_ok:
            flat = self.vector_storage.index
            q_vec = np.asarray(pending[0].q_vec, dtype="float32").reshape(-1) if pending else None
            if q_vec is None:
                q_vec = np.asarray(self.embedding_manager.embed_query(query), dtype="float32").reshape(-1)
        out = scorer.hybrid_tail(tok, flat, rows[0], rows[1], top_k, 60,
                                 post["threshold"] if in_kernel else None, post["knee"] if in_kernel else False,
                                 q_vec, post["lambda"] if in_kernel else 0.5, post["max_bytes"] if in_kernel else None)
        if out is None:
            return None
        fd, fs, n_kept, order = out
        back = {-1 - j: d for d, j in extra.items()}
        combined = []
        for code, sc in zip(fd.tolist(), fs.tolist()):
            d = doc_list[code] if code >= 0 else back[code]
            combined.append({"doc_id": d, "score": sc, "text": self._doc_text(d)})
        if self.cross_encoder:
            combined = self.rerank_with_cross_encoder(query, combined, normalize_ce=True)
        if not in_kernel:
            return combined, None
        # what search_with_strategy does behind the strategy (:121-129), recomputed here from the same float64
        # scores; the kernel's count must agree (it ran MMR over exactly these entries)
        kept = combined
        if post["knee"] and kept:
            cutoff = detect_knee([r["score"] for r in kept])
            kept = [r for r in kept if r["score"] > cutoff]
        kept = [r for r in kept if r["score"] >= post["threshold"]]
        if len(kept) != n_kept:
            raise RuntimeError("rag_hybrid_rrf_mmr kept %d entries, the host rule keeps %d" % (n_kept, len(kept)))
        return combined, (kept, order)

    def reciprocal_rank_fusion(self, *rankings, k=60) -> Dict[str, float]:
        """rrf[doc] += 1/(k+rank) over every entry of every ranking (each sorted by score, stable);
        the accumulation runs in rag_rrf (csrc/cand.hip)."""
        ordered = [[r["doc_id"] for r in sorted(ranking, key=lambda x: x["score"], reverse=True)]
                   for ranking in rankings]
        return candidates.reciprocal_rank_fusion(ordered, k=k)

    # ------------------------------------------------------------------ cross-encoder hook (model is external)

    def rerank_with_cross_encoder(self, query: str, results: List[Dict[str, Any]], normalize_ce: bool = True):
        if not results:
            return results
        pairs = [(query, r.get("text", r.get("chunk", ""))) for r in results]
        ce = np.array(self.cross_encoder.predict(pairs))
        if normalize_ce:
            lo, hi = (float(ce.min()), float(ce.max())) if ce.size else (0.0, 0.0)
            if hi - lo > 1e-6:
                vals = [(s - lo) / (hi - lo) for s in ce]
            else:
                vals = [0.5] * len(ce)
        else:
            vals = [float(s) for s in ce]
        for r, v in zip(results, vals):
            r["ce_score"] = v
        results.sort(key=lambda x: x["ce_score"], reverse=True)
        return results

    # ------------------------------------------------------------------ MMR

    def mmr_selection(self, query: str, results: List[Dict[str, Any]], embedding_manager, lambda_param: float = 0.5,
                      top_k: int = None, max_context_bytes: int = None,
                      reuse_index_vectors: bool = False) -> List[Dict[str, Any]]:
        """Greedy Maximal Marginal Relevance over the candidates' re-encoded texts; the cosine
        matrix and the greedy loop run in rag_mmr (csrc/cand.hip).

        reuse_index_vectors=True (not in the reference; default off because it can
        change results): when every candidate is a chunk the dense index holds, its STORED row is its
        embedding (rag_flat_mmr gathers the rows in HBM) and model.encode (search_engine.py:418) is
        skipped.  Identical to the default whenever the stored rows equal the re-encodings up to a
        positive factor -- MMR normalises -- which holds for rows added through embed_text / add_vector
        and for a saved index.  Document-mode candidates (joined texts, no single row)
        fall back to re-encoding."""
        if not results:
            return results
        top_k = top_k if top_k is not None else len(results)
        texts = [r.get("chunk", r.get("text", "")) for r in results]
        q = np.asarray(embedding_manager.embed_query(query), dtype="float32").reshape(-1)
        byte_lens = [len(t.encode("utf-8")) for t in texts]
        if reuse_index_vectors:
            rows = self._chunk_rows(results)
            if rows is not None:
                index = self.vector_storage.index
                if getattr(index, "sharded", False):
                    # a row-sharded index: the candidates' rows come from the ranks that hold them (one all-gather
                    # of C x d floats), then the same selection kernel on every rank
                    order = candidates.mmr_select(index.gather_rows(rows), q, byte_lens, lambda_param, top_k,
                                                  max_context_bytes)
                else:
                    order = candidates.mmr_select_stored(index, rows, q, byte_lens, lambda_param, top_k,
                                                         max_context_bytes)
                return [results[i] for i in order]
        emb = np.array(embedding_manager.model.encode(texts), dtype="float32")
        order = candidates.mmr_select(emb, q, byte_lens, lambda_param, top_k, max_context_bytes)
        return [results[i] for i in order]

    def _chunk_rows(self, results):
        """Row of the dense index for every candidate, or None if some candidate is not a stored chunk."""
        st = self.vector_storage
        cache = getattr(st, "_chunk_row_map", None)
        if cache is None or cache[0] != len(st.chunks_data):
            table = {}
            for i, cd in enumerate(st.chunks_data):
                table.setdefault((cd["doc_id"], cd["chunk"]), i)
            cache = (len(st.chunks_data), table)
            st._chunk_row_map = cache
        if st.index.ntotal != len(st.chunks_data):
            return None
        rows = []
        for r in results:
            # a chunk hit, or a document entry whose text IS one stored chunk (a one-chunk document: the text
            # :415 hands to model.encode is exactly that chunk's)
            i = cache[1].get((r.get("doc_id"), r.get("chunk", r.get("text"))))
            if i is None:
                return None
            rows.append(i)
        return rows
