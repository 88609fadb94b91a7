"""GPU parity of BM25Okapi / TF-IDF cosine / ranks / RRF / MMR against the CPU oracle.

float64 scoring is compared BIT-EXACT (the kernels perform the same IEEE operations in the
same order as rank_bm25+numpy and scipy); index work (top-k order, ranks, RRF id order) is
exact; MMR (float32 cosines) is compared on the selection order with a near-tie allowance.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _data import zipf_corpus, zipf_queries, word  # noqa: E402
from oracle import fusion as ofu  # noqa: E402
from oracle import sparse as osp  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tfidf_golden.npz")


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("k1,b", [(1.0, 0.65), (1.5, 0.75)])   # config.py:37-38 and rank_bm25 defaults (:305)
def test_bm25_scores_bit_exact_and_topk(engine, k1, b):
    from rag_system_amd.bm25 import BM25Okapi
    corpus = zipf_corpus(3000, 2000, 40, seed=21)
    corpus[100] = list(corpus[7])            # duplicate chunk -> exact ties
    corpus[200] = [word(0)] * 50             # very frequent term only
    ref = osp.BM25OkapiRef(corpus, k1=k1, b=b)
    gpu = BM25Okapi(corpus, k1=k1, b=b)
    assert gpu.corpus_size == ref.corpus_size and gpu.avgdl == ref.avgdl
    assert gpu.average_idf == ref.average_idf
    assert gpu.idf == ref.idf
    queries = zipf_queries(25, 2000, seed=22)
    queries.append([word(0), word(0), word(1)])          # repeated token: added twice
    queries.append(["nonexistenttoken"])                 # all zeros
    queries.append([])                                   # empty query
    queries.append(corpus[7][:12])
    for q in queries:
        s_ref = ref.get_scores(q)
        s_gpu = gpu.get_scores(q)
        assert s_gpu.dtype == np.float64 and s_gpu.shape == s_ref.shape
        assert np.array_equal(_bits(s_gpu), _bits(s_ref)), "BM25 scores are not bit-identical"
        for k in (1, 10, 30, 90, 200):
            ids, sc = gpu.top_k(q, k)
            exp = osp.stable_topk_desc(s_ref, k)
            assert ids.tolist() == exp
            assert np.array_equal(_bits(sc), _bits(s_ref[exp]))


def test_bm25_negative_idf_epsilon_floor(engine):
    # a term present in more than half of the chunks gets a negative idf -> epsilon * average_idf
    from rag_system_amd.bm25 import BM25Okapi
    corpus = [["common", "alpha"], ["common", "beta"], ["common", "gamma", "gamma"], ["delta"], ["common"]]
    ref = osp.BM25OkapiRef(corpus)
    gpu = BM25Okapi(corpus)
    assert ref.idf["common"] == 0.25 * ref.average_idf
    assert gpu.idf == ref.idf
    for q in (["common"], ["common", "gamma"], ["delta", "common", "common"]):
        assert np.array_equal(_bits(gpu.get_scores(q)), _bits(ref.get_scores(q)))
        ids, _ = gpu.top_k(q, 10)   # k > corpus size -> clipped like [:top_k]
        assert ids.tolist() == osp.stable_topk_desc(ref.get_scores(q), 10)


def test_bm25_integer_corpus_builder_matches_token_builder(engine):
    from rag_system_amd.bm25 import BM25Okapi
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 30, 500)
    terms = rng.zipf(1.3, int(lens.sum())) % 300
    indptr = np.concatenate([[0], np.cumsum(lens)])
    docs = [terms[indptr[i]:indptr[i + 1]].tolist() for i in range(500)]
    a = BM25Okapi(docs, k1=1.0, b=0.65)
    bb = BM25Okapi.from_term_ids(indptr, terms, k1=1.0, b=0.65)
    ref = osp.BM25OkapiRef(docs, k1=1.0, b=0.65)
    assert a.idf == ref.idf and bb.idf == ref.idf and bb.avgdl == ref.avgdl
    for q in ([1, 2, 3], [7, 7, 250], [299, 0]):
        s = ref.get_scores(q)
        assert np.array_equal(_bits(a.get_scores(q)), _bits(s))
        assert np.array_equal(_bits(bb.get_scores(q)), _bits(s))


def test_tfidf_golden_fixture_bit_exact(engine):
    """tests/golden/tfidf_golden.npz was produced by scikit-learn's TfidfVectorizer +
    cosine_similarity (make_tfidf_golden.py)."""
    from scipy.sparse import csr_matrix
    from rag_system_amd.tfidf import TfidfScorer
    g = np.load(GOLDEN)
    M = csr_matrix((g["data"], g["indices"], g["indptr"]), shape=tuple(g["shape"]))
    sc = TfidfScorer(None, M)
    for i in range(int(g["n_queries"])):
        got = sc.scores_from_row(g["q%d_indices" % i], g["q%d_data" % i])
        exp = g["q%d_scores" % i]
        assert np.array_equal(_bits(got), _bits(exp)), "query %d: TF-IDF cosine not bit-identical to sklearn" % i


def test_tfidf_live_sklearn_topk_and_ranks(engine):
    from sklearn.feature_extraction.text import TfidfVectorizer
    from rag_system_amd.tfidf import TfidfScorer
    docs = [" ".join(t) for t in zipf_corpus(1200, 3000, 80, seed=31)]
    docs[40] = docs[2]
    vec = TfidfVectorizer()
    M = vec.fit_transform(docs)
    sc = TfidfScorer(vec, M)
    for q in [" ".join(t) for t in zipf_queries(10, 3000, seed=32)] + ["qqqq zzzz", docs[2]]:
        exp = osp.tfidf_cosine(vec, M, q)
        got = sc.scores(q)
        assert np.array_equal(_bits(got), _bits(exp))
        for k in (1, 10, 70):
            ids, s = sc.top_k(q, k)
            order = osp.stable_topk_desc(exp, k)
            assert ids.tolist() == order and np.array_equal(_bits(s), _bits(exp[order]))
        ranks_ref = osp.stable_ranks_desc(exp)
        rows = np.array([0, 2, 40, 41, 1199, 600, 2], dtype=np.int64)
        assert sc.ranks(rows).tolist() == ranks_ref[rows].tolist()
        # the one-call form (rag_tfidf_query) returns what the three calls return
        for k in (0, 10, 70):
            ids, s, rk, rs = sc.query(q, k, rows)
            order = osp.stable_topk_desc(exp, k)
            assert list(ids) == order and np.array_equal(_bits(np.asarray(s)), _bits(exp[order]))
            assert list(rk) == ranks_ref[rows].tolist()
            assert np.array_equal(_bits(np.asarray(rs)), _bits(exp[rows]))
        ids, s, rk, rs = sc.query(q, 5, [])
        assert list(ids) == osp.stable_topk_desc(exp, 5) and len(rk) == 0 and len(rs) == 0


def test_sparse_argument_validation(engine):
    from rag_system_amd.bm25 import make_csc
    with pytest.raises(ValueError):
        make_csc(2, 3, np.array([0, 1, 2]), np.array([0, 5], np.int32), np.array([1.0, 1.0]))  # row out of range
    h = make_csc(2, 3, np.array([0, 1, 2]), np.array([0, 2], np.int32), np.array([1.0, 1.0]))
    with pytest.raises(ValueError):
        h.bm25_scores([0])                      # rag_bm25_set not called
    with pytest.raises(ValueError):
        h.tfidf_scores([7], [1.0])              # term id out of range
    with pytest.raises(ValueError):
        h.ranks([0])                            # no score vector yet


def test_rrf_matches_reference_restatement(engine):
    from rag_system_amd.candidates import reciprocal_rank_fusion
    rng = np.random.default_rng(41)
    for trial in range(20):
        pool = ["doc%d" % i for i in range(int(rng.integers(1, 80)))]
        rankings = []
        for _r in range(3):
            L = int(rng.integers(0, 90))
            docs = [pool[int(i)] for i in rng.integers(0, len(pool), L)]   # repeats allowed (bm25 ranking)
            scores = np.round(rng.random(L), 2)                               # rounded -> score ties
            rankings.append(list(zip(docs, scores.tolist())))
        exp = ofu.reciprocal_rank_fusion(rankings, k=60)
        sorted_ids = [[d for d, _ in sorted(r, key=lambda x: x[1], reverse=True)] for r in rankings]
        got = reciprocal_rank_fusion(sorted_ids, k=60)
        assert list(got.keys()) == list(exp.keys())
        assert [_bits(v) for v in got.values()] == [_bits(v) for v in exp.values()]
    assert reciprocal_rank_fusion([[], []]) == {}


def test_rrf_large_candidate_sets(engine):
    """More than 2048 ranking entries (top_k in the hundreds) take the hash + sort kernel (rrf_sort_kernel):
    same dictionary, same order, same float64 bits; few distinct ids with long runs, all ids distinct, a single
    id, the 16384-entry limit, and the limit exceeded."""
    import time
    from rag_system_amd.candidates import reciprocal_rank_fusion
    from rag_system_amd._lib import EngineError
    rng = np.random.default_rng(43)
    for npool, lens in [(5000, (1500, 1400, 1300)), (40, (900, 800, 700)), (100000, (5000, 5000, 5000)),
                        (1, (3000,)), (3000, (16384,)), (9000, (5461, 5461, 5462)), (2049, (2049,)), (700, (2000, 49))]:
        rankings = [[int(i) for i in rng.integers(0, npool, L)] for L in lens]
        exp = ofu.reciprocal_rank_fusion([[(d, 0.0) for d in r] for r in rankings], k=60)
        t0 = time.perf_counter()
        got = reciprocal_rank_fusion(rankings, k=60)
        ms = (time.perf_counter() - t0) * 1e3
        assert list(got.keys()) == list(exp.keys()), (npool, lens)
        assert [_bits(v) for v in got.values()] == [_bits(v) for v in exp.values()], (npool, lens)
        print("rrf T=%d distinct=%d: %.2f ms" % (sum(lens), len(exp), ms))
    with pytest.raises((EngineError, ValueError)):
        reciprocal_rank_fusion([list(range(16385))], k=60)


def test_mmr_selection_matches_reference_restatement(engine):
    """search_engine.py:396-468 incl. the max_context_bytes branch (:457-463: a pick the budget rejects
    is dropped without being selected, and the loop goes on).  A selection that differs from the
    oracle's must be justified the same way with and without a budget: one of the argmax steps that
    lead to the first differing position (rejected picks included) is a float32 near-tie in float64."""
    from rag_system_amd.candidates import mmr_select
    rng = np.random.default_rng(51)
    n_cases = n_div = n_reject_steps = n_budget_cut = 0
    for trial in range(12):
        C = int(rng.integers(1, 40))
        d = int(rng.choice([8, 384, 768]))
        base = rng.standard_normal((max(1, C // 3), d))
        E = (base[rng.integers(0, len(base), C)] + 0.3 * rng.standard_normal((C, d))).astype(np.float32)
        if C > 3:
            E[2] = E[1]          # exact duplicate candidates
            E[3] = 0.0           # zero vector (norm -> 1e-9 rule)
        q = rng.standard_normal(d).astype(np.float32)
        q /= np.linalg.norm(q)
        blen = rng.integers(10, 2000, C).astype(np.int64)
        for lam, top_k, budget in [(0.5, None, None), (0.85, 5, None), (0.7, 10, 3000), (0.0, 4, None), (1.0, 6, 1500),
                                   (0.6, None, 2500), (0.85, 8, 0), (0.5, 12, 10 ** 9)]:
            exp, trace = ofu.mmr_selection(E, q, blen.tolist(), lam, top_k, budget, return_trace=True)
            got = mmr_select(E, q, blen, lam, top_k, budget)
            n_cases += 1
            rejects = sum(1 for _n, _p, ok in trace if not ok)
            n_reject_steps += rejects
            n_budget_cut += 1 if rejects else 0
            # invariants of the budget rule whatever the tie-breaking: first pick unchecked, then the sum stays in budget
            assert len(set(got)) == len(got) and len(got) <= (top_k if top_k is not None else C)
            if budget is not None and len(got) > 1:
                assert int(blen[got].sum()) <= max(budget, int(blen[got[0]]))
            if got == exp:
                continue
            n_div += 1
            m = min(len(got), len(exp))
            j = next((i for i in range(m) if got[i] != exp[i]), m)
            gaps = ofu.mmr_trace_margins(E, q, lam, trace)
            leading = [abs(g) for (nsel, _p, _ok), g in zip(trace, gaps) if nsel == j]
            assert leading and min(leading) < 1e-5, (
                "MMR diverges from the oracle beyond float32 noise at position %d: %s vs %s (lam=%s top_k=%s "
                "budget=%s, float64 gaps of the steps leading there: %s)" % (j, got, exp, lam, top_k, budget, leading))
    assert n_reject_steps >= 20 and n_budget_cut >= 8, "the byte-budget branch was not exercised"
    assert n_div <= n_cases // 10, "%d of %d selections differ from the oracle" % (n_div, n_cases)


def test_mmr_byte_budget_hand_case(engine):
    """Hand-checkable budget case: orthogonal candidates (no diversity penalty differences), scores
    descending by index, lambda = 1 -> picks go 0,1,2,... and the budget decides alone."""
    from rag_system_amd.candidates import mmr_select
    d = 8
    q = np.zeros(d, np.float32)
    q[0] = 1.0
    E = np.zeros((6, d), np.float32)
    for i in range(6):
        E[i, 0] = 1.0 - 0.1 * i
        E[i, i + 1] = 0.2 + 0.1 * i       # cos to q strictly decreasing with i
    blen = np.array([500, 400, 300, 50, 700, 60], np.int64)
    # first pick (0: 500 B) is never checked; 1 (400) fits 1000; 2 (300) does not -> dropped; 3 (50) fits; 4 no; 5 no (60 > 50 left)
    assert mmr_select(E, q, blen, 1.0, 6, 1000) == [0, 1, 3]
    assert ofu.mmr_selection(E, q, blen.tolist(), 1.0, 6, 1000) == [0, 1, 3]
    assert mmr_select(E, q, blen, 1.0, 6, 100) == [0]              # budget below the first pick: it still goes in
    assert mmr_select(E, q, blen, 1.0, 6, 1010) == [0, 1, 3, 5]
    assert mmr_select(E, q, blen, 1.0, 2, 10 ** 6) == [0, 1]
    assert mmr_select(E, q, blen, 1.0, None, None) == [0, 1, 2, 3, 4, 5]


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_mmr_on_stored_rows_equals_mmr_on_uploaded_rows(engine, storage):
    """rag_flat_mmr (candidates = rows of the index, gathered in HBM) against rag_mmr on the same values
    handed over as host rows: the same kernel body, so the selections are identical."""
    from oracle import dense as od
    from rag_system_amd import faiss_compat as faiss
    from rag_system_amd.candidates import mmr_select, mmr_select_stored
    rng = np.random.default_rng(88)
    n, d = 6000, 100                      # d = 100: padded rows (ld = 100 / 104)
    X = (rng.standard_normal((n, d)) * rng.uniform(0.2, 5.0, (n, 1))).astype(np.float32)
    X[17] = 0.0
    idx = faiss.IndexFlat(d, 0, storage=storage)
    idx.add(X)
    Xs = od.bf16_round(X) if storage == "bf16" else X
    q = rng.standard_normal(d).astype(np.float32)
    for C, lam, top_k, budget in [(1, 0.5, None, None), (12, 0.7, 10, None), (40, 0.3, 15, 5000), (90, 1.0, 30, None),
                                  (30, 0.0, 30, 100)]:
        rows = rng.choice(n, C, replace=False).astype(np.int64)
        if C > 5:
            rows[3] = 17                  # a zero row (norm -> 1e-9 rule)
            rows[4] = rows[2]             # the same chunk twice
        blen = rng.integers(10, 900, C).astype(np.int64)
        exp = mmr_select(Xs[rows], q, blen, lam, top_k, budget)
        got = mmr_select_stored(idx, rows, q, blen, lam, top_k, budget)
        assert got == exp, (storage, C, lam, top_k, budget)
    with pytest.raises(ValueError):
        mmr_select_stored(idx, [n], q, [1])


def test_csc_large_single_term_and_many_rows(engine):
    """1M rows, a term present in every row plus rare terms: ordering among equal scores and
    the multi-block top-k / rank paths at BASELINE C5 scale."""
    from rag_system_amd.bm25 import make_csc
    n = 1_000_000
    rng = np.random.default_rng(61)
    rare = np.sort(rng.choice(n, 5000, replace=False)).astype(np.int32)
    indptr = np.array([0, n, n + 5000], dtype=np.int64)
    rows = np.concatenate([np.arange(n, dtype=np.int32), rare])
    vals = np.concatenate([np.ones(n), rng.integers(1, 5, 5000).astype(np.float64)])
    h = make_csc(2, n, indptr, rows, vals)
    s, D, I = h.tfidf_scores([0, 1], [0.5, 0.25], want_scores=True, k=100)
    exp = np.full(n, 0.5)
    exp[rare] = 0.5 + 0.25 * vals[n:]
    assert np.array_equal(s, exp)
    order = np.lexsort((np.arange(n), -exp))[:100]
    assert I.tolist() == order.tolist() and np.array_equal(D, exp[order])
    q = np.array([0, int(rare[0]), n - 1, int(rare[-1]), 12345], dtype=np.int64)
    full_rank = np.empty(n, dtype=np.int64)
    full_rank[np.lexsort((np.arange(n), -exp))] = np.arange(n)
    assert h.ranks(q).tolist() == full_rank[q].tolist()
    # one call, 1024 queried rows (its limit), ranks only / ranks and scores
    q2 = np.concatenate([q, rng.integers(0, n, 1019)]).astype(np.int64)
    D2, I2, rk, rs = h.tfidf_query([0, 1], [0.5, 0.25], 100, q2)
    assert I2.tolist() == order.tolist() and np.array_equal(D2, exp[order])
    assert rk.tolist() == full_rank[q2].tolist() and np.array_equal(rs, exp[q2])
    D3, I3, rk3, rs3 = h.tfidf_query([0, 1], [0.5, 0.25], 10, q2[:7], want_scores=False)
    assert rs3 is None and rk3.tolist() == full_rank[q2[:7]].tolist() and I3.tolist() == order[:10].tolist()
    with pytest.raises(ValueError):
        h.tfidf_query([0], [1.0], 10, np.zeros(1025, dtype=np.int64))
    with pytest.raises(ValueError):
        h.tfidf_query([0], [1.0], 10, np.array([n], dtype=np.int64))


def test_tfidf_query_in_two_halves_equals_one_call(engine):
    """rag_tfidf_query_begin / _end: the scoring enqueued on the handle's own stream, the head of the ranking
    and the rank query delivered later -- bit-identical to rag_tfidf_query, also with a smaller k at the end,
    beside a BM25 scoring on another handle, with 1024 queried rows, after an abort; the handle refuses other
    calls while a query is begun."""
    from rag_system_amd import _lib
    from rag_system_amd.bm25 import make_csc
    lib = _lib.engine()
    rng = np.random.default_rng(67)
    n, nt = 300_000, 50
    df = rng.integers(200, 60_000, nt)
    indptr = np.concatenate([[0], np.cumsum(df)]).astype(np.int64)
    rows = np.concatenate([np.sort(rng.choice(n, int(c), replace=False)) for c in df]).astype(np.int32)
    vals = rng.random(len(rows)) + 0.01
    h = make_csc(nt, n, indptr, rows, vals)
    other = make_csc(nt, n, indptr, rows, vals)
    n_split = 0
    for trial in range(12):
        t = rng.choice(nt, int(rng.integers(1, 8)), replace=False).astype(np.int32)
        w = rng.random(len(t))
        q = rng.integers(0, n, int(rng.choice([1, 37, 256, 1024]))).astype(np.int64)
        q[-1] = q[0]                                   # a row queried twice
        kb = int(rng.choice([10, 70, 200]))
        k = int(rng.integers(1, kb + 1))
        D0, I0, rk0, rs0 = h.tfidf_query(t, w, k, q)
        if not h.tfidf_query_begin(t, w, kb):
            assert kb == 200                          # more than the one-launch selection serves here: nothing begun
            h.tfidf_query(t, w, k, q)
            continue
        n_split += 1
        with pytest.raises(Exception):
            h.tfidf_query(t, w, k, q)                 # refused while begun
        with pytest.raises(Exception):
            h.tfidf_query_begin(t, w, kb)
        other.tfidf_scores(t, w, want_scores=False, k=5)   # another handle works beside it (library stream)
        if trial % 3 == 2:       # the three rank kernels instead of the one-launch rank kernel (<= 256 rows)
            _lib.check(lib.rag_set_option(b"sparse_rank_one", -1))
        try:
            D1, I1, rk1, rs1 = h.tfidf_query_end(k, q)
        finally:
            _lib.check(lib.rag_set_option(b"sparse_rank_one", 0))
        assert I1.tolist() == I0.tolist() and np.array_equal(_bits(D1), _bits(D0))
        assert rk1.tolist() == rk0.tolist() and np.array_equal(_bits(rs1), _bits(rs0))
    assert n_split >= 3
    with pytest.raises(Exception):
        h.tfidf_query_end(5, q)                       # nothing begun
    assert h.tfidf_query_begin(t, w, 50)
    with pytest.raises(Exception):
        h.tfidf_query_end(51, q)                      # beyond the bound: refused, and the query is over
    assert h.tfidf_query_begin(t, w, 50)
    h.tfidf_query_abort()
    D2, I2, rk2, rs2 = h.tfidf_query(t, w, k, q)      # the handle is usable again
    assert I2.tolist() == I0.tolist() and rk2.tolist() == rk0.tolist()


def _topk_of_vector(h_cache, scores, k):
    """Top-k of an arbitrary float64 vector through the TF-IDF entry point: one term present in
    every row with val = the vector and query weight 1.0 gives score == val bit for bit."""
    from rag_system_amd.bm25 import make_csc
    n = len(scores)
    h = make_csc(1, n, np.array([0, n], dtype=np.int64), np.arange(n, dtype=np.int32),
                 np.ascontiguousarray(scores, dtype=np.float64))
    s, D, I = h.tfidf_scores([0], [1.0], want_scores=True, k=k)
    assert np.array_equal(_bits(s), _bits(scores))
    return D, I


def test_topk_threshold_path_matches_stable_sort(engine):
    """Large score vectors take the threshold path of the top-k (span maxima -> k-th largest ->
    append scores above it -> exact selection, ties completed by ascending row).  Checked
    against the stable descending sort the reference takes (vector_storage.py:180), including
    vectors that force the fallback to the list kernels, and against the list kernels alone."""
    from rag_system_amd import _lib
    lib = _lib.engine()
    rng = np.random.default_rng(77)
    n = 300_007
    cases = {
        "random": rng.standard_normal(n),
        "few_positive": np.where(rng.random(n) < 0.0005, rng.random(n) * 7.0, 0.0),       # BM25 with rare terms
        "ties_everywhere": rng.integers(0, 6, n).astype(np.float64),                       # 6 distinct values
        "all_equal": np.full(n, 0.125),
        "descending_by_row": np.linspace(5.0, -5.0, n),                                    # overflow -> list kernels
        "ascending_by_row": np.linspace(-5.0, 5.0, n),
        "one_span_holds_the_top": np.concatenate([rng.random(n - 200), 10.0 + rng.random(200)]),
        "boundary_tie_far_away": np.concatenate([[3.0] * 50, rng.random(n - 52), [3.0, 3.0]]),
    }
    spread = rng.standard_normal(n)
    spread[np.linspace(1000, n - 1000, 8).astype(np.int64)] = 4.0    # 8 equal scores far apart ...
    cases["boundary_tie_spread"] = spread                             # ... around rank 10-20: k = 12 cuts the group
    for name, sc in cases.items():
        order = np.lexsort((np.arange(n), -sc))
        for k in (1, 10, 12, 100, 256):
            D, I = _topk_of_vector(None, sc, k)
            assert I.tolist() == order[:k].tolist(), (name, k)
            assert np.array_equal(_bits(D), _bits(sc[order[:k]])), (name, k)
    try:
        _lib.check(lib.rag_set_option(b"sparse_topk_lists", 1))
        D, I = _topk_of_vector(None, cases["random"], 100)
        order = np.lexsort((np.arange(n), -cases["random"]))
        assert I.tolist() == order[:100].tolist()
    finally:
        lib.rag_set_option(b"sparse_topk_lists", 0)


def test_count_ahead_of_arbitrary_pairs(engine):
    """rag_csc_count_ahead: rows of the last score vector ahead of (score, id) pairs that need not be rows of the
    index -- what a row-sharded TF-IDF / BM25 index sums over its shards to rank a document.  Against numpy, with
    exact score ties, ids below 0 and beyond n (another shard's rows), own rows (= rag_csc_ranks), repeated pairs,
    and more pairs than one launch takes."""
    from scipy.sparse import random as sprandom
    from rag_system_amd.tfidf import TfidfScorer
    rng = np.random.default_rng(17)
    n, V = 30_000, 500
    M = sprandom(n, V, density=0.02, format="csr", random_state=3, dtype=np.float64)
    M.data = np.round(M.data * 4) / 4 + 0.25
    ts = TfidfScorer(None, M)
    s = ts.scores_from_row(np.array([3, 17, 40], dtype=np.int32), np.array([1.0, 2.0, 1.0]))
    j = np.arange(n)
    own = rng.choice(n, 300, replace=False)
    keys = np.concatenate([s[own], s[rng.choice(n, 900)], np.array([0.0, -1.0, 10.0, s.max()])])
    ids = np.concatenate([own, rng.integers(-50_000, 80_000, 900), np.array([-5, 10, n + 7, -1])]).astype(np.int64)
    keys = np.concatenate([keys, keys[:5]])
    ids = np.concatenate([ids, ids[:5]])
    got = ts.count_ahead(keys, ids)
    exp = np.array([int(((s > k) | ((s == k) & (j < i))).sum()) for k, i in zip(keys, ids)])
    assert got.tolist() == exp.tolist()
    assert got[:300].tolist() == ts.ranks(own).tolist()
    assert len(keys) > 1024
