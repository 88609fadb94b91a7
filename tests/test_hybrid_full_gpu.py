"""BASELINE config C5 at full size (1M documents, 200k-term Zipf vocabulary, ~80M postings, dense
1M x 384): every stage of strategy=rrf + MMR through the engine, checked against oracle/ --
BM25 ids and float64 scores bit-exact, TF-IDF ids / ranks exact, dense ids exact, fused top_k
identical to RRF over the three FULL rankings (search_engine.py:352-359), MMR order identical.
The same checks benchmarks/bench_hybrid.py runs after its timed loop."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_c5_rrf_mmr_stage_by_stage_at_1m_documents(engine):
    from benchmarks.hybrid_workload import HybridWorkload
    nq = 5
    wl = HybridWorkload(1_000_000, 200_000, 384, nq, top_k=10)
    assert wl.nnz > 60_000_000
    # the three-call route (rag_rrf, rag_tfidf_query_end, rag_flat_mmr): every stage visible and checked
    results = [wl.query(qi, tail=False) for qi in range(nq)]
    for r in results:
        assert len(r["fused"]) == 10 and 1 <= len(r["mmr"]) <= 10 and len(r["bm25"][0]) == 30
        assert r["tfidf"] is not None and len(r["ranks"]) == len(r["cand"])
    assert wl.check(list(range(nq)), results) == nq
    # the one-call tail (rag_hybrid_rrf_mmr, the default route): the same fused list, scores bit for bit, and
    # the same MMR order -- and checked against oracle/ on its own
    assert wl.tail_ok
    fused = [wl.query(qi) for qi in range(nq)]
    for a, b in zip(results, fused):
        assert b["tfidf"] is None, "the one-call tail was not taken"
        assert a["fused"] == b["fused"] and a["mmr"] == b["mmr"] and a["dense"] == b["dense"]
        assert a["bm25"][0] == b["bm25"][0]
    assert wl.check(list(range(nq)), fused) == nq
