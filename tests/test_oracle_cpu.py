"""CPU tests: the oracle against independent restatements and the golden fixtures.

The reference holds no test, fixture or golden vector for faiss / rank_bm25, so the
dense and BM25 oracles are checked here against independent numpy / hand-computed results;
the TF-IDF oracle IS scikit-learn and the committed fixture is re-derived from it.
"""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _data import zipf_corpus, zipf_queries  # noqa: E402
from oracle import dense as od  # noqa: E402
from oracle import fusion as ofu  # noqa: E402
from oracle import sparse as osp  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tfidf_golden.npz")
FMAX = np.finfo(np.float32).max


def _numpy_flat(X, Q, k, metric):
    X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
    if metric == 0:
        S = Q64 @ X64.T
        key = -S
    else:
        S = ((Q64[:, None, :] - X64[None, :, :]) ** 2).sum(-1)
        key = S
    I = np.full((len(Q), k), -1, dtype=np.int64)
    D = np.full((len(Q), k), -FMAX if metric == 0 else FMAX, dtype=np.float64)
    for qi in range(len(Q)):
        order = np.lexsort((np.arange(len(X)), key[qi]))[:k]
        I[qi, :len(order)] = order
        D[qi, :len(order)] = S[qi, order]
    return D, I


@pytest.mark.parametrize("metric", [0, 1])
def test_dense_oracle_matches_numpy(metric):
    rng = np.random.default_rng(0)
    for n, d, nq, k in [(1, 8, 1, 3), (50, 37, 4, 10), (2000, 384, 3, 30), (17, 4, 2, 40)]:
        X = rng.standard_normal((n, d)).astype(np.float32)
        X[n // 2] = X[0]  # exact duplicate row
        Q = rng.standard_normal((nq, d)).astype(np.float32)
        D, I, D64 = od.flat_search(X, Q, k, metric)
        Dn, In = _numpy_flat(X, Q, k, metric)
        od.check_topk(D, I, Dn, In, 1e-6, 1e-9)
        assert np.allclose(D64, Dn, rtol=1e-12, atol=1e-12)
        # fp32-accumulating variant (faiss' own precision) agrees within the parity tolerance
        D32, I32, _ = od.flat_search(X, Q, k, metric, accum64=False)
        od.check_topk(D32, I32, Dn, In, 1e-5, 1e-5 * d)


def test_dense_oracle_tie_rule_and_padding():
    X = np.ones((6, 4), np.float32)
    Q = np.ones((1, 4), np.float32)
    D, I, _ = od.flat_search(X, Q, 8, 0)
    assert I[0].tolist() == [0, 1, 2, 3, 4, 5, -1, -1]
    assert D[0, :6].tolist() == [4.0] * 6 and (D[0, 6:] == -FMAX).all()
    D, I, _ = od.flat_search(X, Q, 8, 1)
    assert I[0].tolist() == [0, 1, 2, 3, 4, 5, -1, -1] and (D[0, 6:] == FMAX).all()
    D, I, _ = od.flat_search(np.zeros((0, 4), np.float32), Q, 2, 0)
    assert (I == -1).all()


def test_synthetic_generator_properties():
    a = od.synth_rows(1234, 0, 200, 384)
    b = od.synth_rows(1234, 100, 100, 384)
    assert a[100:].tobytes() == b.tobytes(), "rows must depend on (seed, global row) only"
    assert not np.array_equal(a, od.synth_rows(1235, 0, 200, 384))
    nrm = np.linalg.norm(a.astype(np.float64), axis=1)
    assert np.allclose(nrm, 1.0, atol=1e-6)
    assert abs(a.mean()) < 1e-3 and abs(a.std() - 1 / math.sqrt(384)) < 1e-3
    # streaming search == stored search
    Q = od.synth_queries(4321, 3, 384)
    D1, I1, E1 = od.flat_search(a, Q, 10, 0)
    D2, I2, E2 = od.flat_search_synth(1234, 0, 200, 384, Q, 10, 0)
    assert np.array_equal(I1, I2) and np.array_equal(E1, E2)


def test_merge_oracle_equals_whole_search():
    rng = np.random.default_rng(3)
    X = rng.standard_normal((900, 16)).astype(np.float32)
    X[700] = X[3]
    Q = rng.standard_normal((4, 16)).astype(np.float32)
    for metric in (0, 1):
        Dw, Iw, _ = od.flat_search(X, Q, 12, metric)
        parts = [(0, 300), (300, 301), (301, 900)]
        Dl, Il = [], []
        for lo, hi in parts:
            D, I, _ = od.flat_search(X[lo:hi], Q, 12, metric)
            Dl.append(D)
            Il.append(np.where(I >= 0, I + lo, -1))
        Dm, Im = od.topk_merge(metric, np.stack(Dl), np.stack(Il))
        assert np.array_equal(Im, Iw) and np.array_equal(Dm, Dw)


def test_cpu_baseline_scan_agrees_with_oracle():
    X = od.synth_rows(1, 0, 5000, 384)
    Q = od.synth_queries(2, 1, 384)
    _, Ir, D64 = od.flat_search(X, Q, 10, 0)
    for nt in (1, 4):
        D, I = od.cpu_baseline_search(X, Q, 10, nt)
        od.check_topk(D, I, D64, Ir, 1e-5, 2e-6)


def test_bm25_oracle_hand_computed():
    corpus = [["a", "b", "a"], ["b", "c"], ["c", "c", "c", "d"]]
    m = osp.BM25OkapiRef(corpus, k1=1.5, b=0.75, epsilon=0.25)
    assert m.avgdl == 3.0 and m.doc_len == [3, 2, 4]
    idf_a = math.log(3 - 1 + 0.5) - math.log(1 + 0.5)
    idf_b = math.log(3 - 2 + 0.5) - math.log(2 + 0.5)          # negative
    idf_c = idf_b
    idf_d = idf_a
    avg = (idf_a + idf_b + idf_c + idf_d) / 4
    assert m.idf["a"] == idf_a and m.idf["b"] == 0.25 * avg and m.idf["c"] == 0.25 * avg
    s = m.get_scores(["a", "c", "zzz"])
    def term(idf, tf, dl):
        return idf * (tf * 2.5 / (tf + 1.5 * (1 - 0.75 + 0.75 * dl / 3.0)))
    exp = [term(idf_a, 2, 3), term(0.25 * avg, 1, 2), term(0.25 * avg, 3, 4)]
    assert np.allclose(s, exp, rtol=0, atol=1e-15)
    assert osp.stable_topk_desc([1.0, 3.0, 3.0, 0.0], 3) == [1, 2, 0]
    assert osp.stable_ranks_desc([1.0, 3.0, 3.0, 0.0]).tolist() == [2, 0, 1, 3]


def test_tfidf_golden_reproducible_with_installed_sklearn():
    """The fixture is data produced by scikit-learn; the installed scikit-learn must still
    produce it (pins the fixture to the library the reference calls)."""
    from scipy.sparse import csr_matrix
    from sklearn.metrics.pairwise import cosine_similarity
    g = np.load(GOLDEN)
    M = csr_matrix((g["data"], g["indices"], g["indptr"]), shape=tuple(g["shape"]))
    for i in range(int(g["n_queries"])):
        idx, dat = g["q%d_indices" % i], g["q%d_data" % i]
        q = csr_matrix((dat, idx, np.array([0, len(idx)])), shape=(1, M.shape[1]))
        s = cosine_similarity(q, M)[0]
        assert np.array_equal(s, g["q%d_scores" % i])
    # properties the GPU path relies on: rows are unit-norm, idf formula, alphabetical vocabulary
    from sklearn.feature_extraction.text import TfidfVectorizer
    docs = [" ".join(t) for t in zipf_corpus(50, 200, 30, seed=1)]
    v = TfidfVectorizer()
    X = v.fit_transform(docs)
    assert np.allclose(np.sqrt(X.multiply(X).sum(axis=1)), 1.0)
    names = v.get_feature_names_out().tolist()
    assert names == sorted(names)
    df = np.asarray((X > 0).sum(axis=0)).ravel()
    assert np.allclose(v.idf_, np.log((1 + 50) / (1 + df)) + 1)


def test_fusion_oracle_small_cases():
    r = ofu.reciprocal_rank_fusion([[("a", 0.9), ("b", 0.5)], [("b", 3.0), ("b", 2.0), ("c", 1.0)], []], k=60)
    assert list(r.keys()) == ["a", "b", "c"]
    assert r["a"] == 1 / 60 and r["b"] == 1 / 61 + 1 / 60 + 1 / 61 and r["c"] == 1 / 62
    assert ofu.detect_knee([0.9, 0.8, 0.3, 0.2]) == 0.3
    assert ofu.detect_knee([0.5]) == 0.5 and ofu.detect_knee([]) == 0.0
    E = np.eye(4, dtype=np.float32)
    E[1] = E[0]
    q = np.array([1, 0.5, 0.2, 0], np.float32)
    sel = ofu.mmr_selection(E, q, [10, 10, 10, 10], 0.5, 3, None)
    assert sel[0] == 0 and sel[1] == 2          # the duplicate of the first pick is penalised
    sel = ofu.mmr_selection(E, q, [10, 100, 10, 10], 0.5, 4, 25)
    assert sel == [0, 2]                          # byte budget skips candidates without selecting


def test_bf16_rounding_matches_torch_cast():
    """oracle bf16_rne (shared bit-for-bit by the HIP engine's storage rounding) against an
    independent implementation: torch's float32 -> bfloat16 cast (round-to-nearest-even)."""
    import torch
    rng = np.random.default_rng(11)
    a = (rng.standard_normal(200_000) * rng.choice([1e-30, 1e-3, 1.0, 1e20], size=200_000)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1e-40, -1e-45, np.inf, -np.inf,
                        3.3895314e38, 65504.0], dtype=np.float32)
    a = np.concatenate([a, special])
    ref = torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
    got = od.bf16_round(a)
    assert got.tobytes() == ref.tobytes()
    assert np.isnan(od.bf16_round(np.array([np.nan], dtype=np.float32))[0])


def test_postings_form_of_bm25_equals_dict_form():
    """oracle.sparse.bm25_scores_postings (vectorised, used at 1M documents) is bit-identical to
    the rank_bm25 restatement BM25OkapiRef on a corpus both can hold."""
    docs = zipf_corpus(400, 300, 60, seed=3)
    ref = osp.BM25OkapiRef(docs, k1=1.0, b=0.65)
    vocab = {w: i for i, w in enumerate(ref.idf)}
    post = {}
    for di, d in enumerate(docs):
        for w, f in ref.doc_freqs[di].items():
            post.setdefault(vocab[w], []).append((di, f))
    indptr = np.zeros(len(vocab) + 1, np.int64)
    rows, tf = [], []
    for t in range(len(vocab)):
        indptr[t + 1] = indptr[t] + len(post[t])
        rows += [p[0] for p in post[t]]
        tf += [float(p[1]) for p in post[t]]
    idf = np.array([ref.idf[w] for w in vocab])
    for q in zipf_queries(8, 300, seed=4):
        tids = [vocab[w] for w in q if w in vocab]
        got = osp.bm25_scores_postings(indptr, np.array(rows), np.array(tf), idf, ref.doc_len, ref.avgdl, 1.0, 0.65, tids)
        assert got.tobytes() == ref.get_scores(q).tobytes()


def test_check_topk_rejects_a_repeated_id():
    """A duplicate id that scores within tolerance of the oracle's entry must not pass as a near-tie."""
    import pytest
    D = np.array([[3.0, 2.0, 2.0]])
    I_ref = np.array([[5, 7, 8]])
    od.check_topk(D, I_ref, D, I_ref)
    with pytest.raises(AssertionError, match="more than once"):
        od.check_topk(D, np.array([[5, 7, 7]]), D, I_ref)
    pad = np.array([[5, -1, -1]])
    Dp = np.array([[3.0, -3.4028234663852886e38, -3.4028234663852886e38]])
    od.check_topk(Dp, pad, Dp, pad)      # pads repeat by construction


def test_fallback_tokenizer_reproduces_nltks_documented_treebank_examples():
    """The regex fallback for nltk.word_tokenize restates the Treebank rules: the three examples nltk's own
    documentation gives for TreebankWordTokenizer come out as documented (no punkt: mid-text sentence periods
    stay attached, as in that tokenizer)."""
    from rag_system_amd.text import _treebank_tokenize as tb
    assert tb("They'll save and invest more.") == ["They", "'ll", "save", "and", "invest", "more", "."]
    assert tb("hi, my name can't hello,") == ["hi", ",", "my", "name", "ca", "n't", "hello", ","]
    assert tb("Good muffins cost $3.88\nin New York.  Please buy me\ntwo of them.\nThanks.") == [
        "Good", "muffins", "cost", "$", "3.88", "in", "New", "York.", "Please", "buy", "me", "two", "of", "them.",
        "Thanks", "."]
    assert tb('he said "no" (twice) -- really?') == ["he", "said", "``", "no", "''", "(", "twice", ")", "--", "really", "?"]
    assert tb("") == [] and tb("plain words only") == ["plain", "words", "only"]
