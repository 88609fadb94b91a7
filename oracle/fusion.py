"""CPU restatement of the candidate-set stages -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Restated from the reference source and PINNED (round 3) against the outputs of the reference's own
rag_system/search_engine.py: tests/golden/strategy_golden.json, written in the build container by
tests/golden/make_strategy_golden.py, replayed by tests/test_strategy_golden.py (exact equality):
  reciprocal_rank_fusion   search_engine.py:352-359
  mmr_selection            search_engine.py:396-468 (numeric part; embeddings are inputs)
  detect_knee              search_engine.py:37-48
"""
import numpy as np


def reciprocal_rank_fusion(rankings, k=60):
    """rankings: list of lists of (doc_id, score).  Returns dict doc_id -> score in
    first-insertion order (search_engine.py:352-359)."""
    rrf_scores = {}
    for ranking in rankings:
        sorted_docs = sorted(ranking, key=lambda x: x[1], reverse=True)
        for rank, (doc_id, _s) in enumerate(sorted_docs):
            rrf_scores[doc_id] = rrf_scores.get(doc_id, 0) + 1.0 / (k + rank)
    return rrf_scores


def detect_knee(scores):
    if len(scores) < 2:
        return scores[0] if scores else 0.0
    diffs = [scores[i] - scores[i + 1] for i in range(len(scores) - 1)]
    max_drop_index = diffs.index(max(diffs))
    return scores[max_drop_index + 1]


def mmr_selection(cand_emb, query_vec, byte_lens, lambda_param=0.5, top_k=None, max_context_bytes=None,
                  return_trace=False):
    """search_engine.py:419-465 with the model outputs given: returns selected indices.
    return_trace: also the list of argmax steps (n_selected_before, picked index, accepted) --
    a pick the byte budget rejects (:457-463) is a step too."""
    from sklearn.metrics.pairwise import cosine_similarity
    n = len(cand_emb)
    if n == 0:
        return ([], []) if return_trace else []
    top_k = top_k if top_k is not None else n
    candidate_embeddings = np.array(cand_emb, dtype="float32")
    norms = np.linalg.norm(candidate_embeddings, axis=1, keepdims=True)
    norms[norms == 0] = 1e-9
    candidate_embeddings = candidate_embeddings / norms
    query_vec = np.asarray(query_vec, dtype="float32").reshape(1, -1)
    query_similarities = cosine_similarity(candidate_embeddings, query_vec).flatten()
    selected_indices = []
    candidate_indices = list(range(n))
    first_index = int(np.argmax(query_similarities))
    trace = [(0, first_index, True)]
    selected_indices.append(first_index)
    candidate_indices.remove(first_index)
    total_bytes = byte_lens[first_index]
    while candidate_indices and len(selected_indices) < top_k:
        mmr_scores = []
        for i in candidate_indices:
            sim_to_query = query_similarities[i]
            sim_to_selected = [
                cosine_similarity(candidate_embeddings[i].reshape(1, -1),
                                  candidate_embeddings[j].reshape(1, -1))[0, 0]
                for j in selected_indices
            ]
            max_sim = max(sim_to_selected) if sim_to_selected else 0.0
            score = lambda_param * sim_to_query - (1 - lambda_param) * max_sim
            mmr_scores.append(score)
        best_idx_relative = int(np.argmax(mmr_scores))
        best_idx = candidate_indices[best_idx_relative]
        if max_context_bytes is not None:
            if total_bytes + byte_lens[best_idx] > max_context_bytes:
                trace.append((len(selected_indices), best_idx, False))
                candidate_indices.remove(best_idx)
                continue
            else:
                total_bytes += byte_lens[best_idx]
        trace.append((len(selected_indices), best_idx, True))
        selected_indices.append(best_idx)
        candidate_indices.remove(best_idx)
    return (selected_indices, trace) if return_trace else selected_indices


def mmr_trace_margins(cand_emb, query_vec, lambda_param, trace):
    """float64 gap between the picked candidate's MMR score and the best other live candidate at
    EVERY argmax step of `trace` (budget-rejected picks included), replaying the live / selected
    sets the trace implies.  A selection that differs from the oracle's is only acceptable where
    one of the steps leading to the first differing position has a gap within float32 noise."""
    E = np.asarray(cand_emb, dtype=np.float64)
    nrm = np.linalg.norm(E, axis=1, keepdims=True)
    E = E / np.where(nrm == 0, 1e-9, nrm)
    q = np.asarray(query_vec, dtype=np.float64)
    q = q / max(np.linalg.norm(q), 1e-300)
    simq = E @ q
    G = E @ E.T
    live = list(range(len(E)))
    sel = []
    gaps = []
    for _nsel, pick, accepted in trace:
        if not sel:
            sc = {i: simq[i] for i in live}
        else:
            sc = {i: lambda_param * simq[i] - (1 - lambda_param) * max(G[i, j] for j in sel) for i in live}
        others = [v for i, v in sc.items() if i != pick]
        gaps.append(sc[pick] - max(others) if others else np.inf)
        if accepted:
            sel.append(pick)
        live.remove(pick)
    return gaps
