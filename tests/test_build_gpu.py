"""GPU parity of the device-side postings build (csrc/build.hip, rag_postings_build)
against a numpy restatement of what rank_bm25's BM25Okapi.__init__ counts (per-document term
frequencies, vocabulary in first-appearance order; called at vector_storage.py:170,305).  Integer
work: everything bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _numpy_postings(doc_indptr, words):
    """The pre-device-build host path of BM25Okapi.from_term_ids, kept here as the checker."""
    doc_indptr = np.asarray(doc_indptr, dtype=np.int64)
    words = np.asarray(words, dtype=np.int64)
    n = len(doc_indptr) - 1
    lens = np.diff(doc_indptr)
    doc_of = np.repeat(np.arange(n, dtype=np.int64), lens)
    uniq, first_pos = np.unique(words, return_index=True)
    order = np.argsort(first_pos, kind="stable")
    tid_of_uniq = np.empty(len(uniq), dtype=np.int64)
    tid_of_uniq[order] = np.arange(len(uniq))
    tids = tid_of_uniq[np.searchsorted(uniq, words)]
    key = tids * n + doc_of
    ukey, tf = np.unique(key, return_counts=True)
    p_term = ukey // n
    p_doc = (ukey - p_term * n).astype(np.int32)
    df = np.bincount(p_term, minlength=len(uniq)).astype(np.int64)
    return uniq[order].astype(np.int32), np.concatenate([np.zeros(1, np.int64), np.cumsum(df)]), p_doc, tf.astype(np.float64)


def _corpus(rng, n_docs, n_words, mean_len, empty_frac=0.1):
    lens = rng.poisson(mean_len, n_docs)
    lens[rng.random(n_docs) < empty_frac] = 0
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    # Zipf-like draws: frequent words repeat inside a document, rare ones appear once in the corpus
    w = np.minimum((rng.pareto(1.1, int(indptr[-1])) * 3).astype(np.int64), n_words - 1)
    return indptr, w


@pytest.mark.parametrize("n_docs,n_words,mean_len", [(1, 5, 7), (13, 50, 4), (2_000, 300, 30), (50_000, 20_000, 60),
                                                      (300, 100_000, 3)])
def test_device_postings_equal_numpy(engine, n_docs, n_words, mean_len):
    from rag_system_amd.bm25 import device_postings
    rng = np.random.default_rng(n_docs + n_words)
    indptr, w = _corpus(rng, n_docs, n_words, mean_len)
    if len(w) == 0:
        w = np.zeros(1, np.int64)
        indptr[-1] = 1
    tw, ip, rows, tf = device_postings(indptr, w.astype(np.int32), n_words)
    tw0, ip0, rows0, tf0 = _numpy_postings(indptr, w)
    assert np.array_equal(tw, tw0) and np.array_equal(ip, ip0)
    assert np.array_equal(rows, rows0) and np.array_equal(tf, tf0)
    assert tf.sum() == len(w)


def test_bad_word_and_empty_corpus(engine):
    from rag_system_amd.bm25 import device_postings
    with pytest.raises(ValueError):
        device_postings(np.array([0, 2]), np.array([1, 9], dtype=np.int32), 5)      # word 9 outside [0, 5)
    with pytest.raises(ValueError):
        device_postings(np.array([0, 2]), np.array([0, 0], dtype=np.int32), 0)      # tokens but an empty vocabulary
    # the handle-less call leaves the library usable after a refused build (no copy pending into freed buffers)
    tw, ip, rows, tf = device_postings(np.array([0, 2]), np.array([1, 1], dtype=np.int32), 5)
    assert tw.tolist() == [1] and ip.tolist() == [0, 1] and rows.tolist() == [0] and tf.tolist() == [2.0]
    tw, ip, rows, tf = device_postings(np.array([0, 0, 0]), np.zeros(0, np.int32), 4)  # two empty documents
    assert len(tw) == 0 and ip.tolist() == [0] and len(rows) == 0


def test_bm25_constructors_use_the_device_build_and_match_the_oracle(engine):
    """String corpus and integer corpus: idf dictionary (order included), avgdl, doc_len and scores equal
    the rank_bm25 restatement of oracle/sparse.py bit for bit."""
    from oracle import sparse as osp
    from rag_system_amd.bm25 import BM25Okapi
    rng = np.random.default_rng(5)
    vocab = ["w%d" % i for i in range(400)]
    corpus = [[vocab[min(int(x), 399)] for x in rng.pareto(1.0, rng.integers(0, 40)) * 4] for _ in range(3000)]
    corpus[7] = []
    for k1, b in ((1.5, 0.75), (1.0, 0.65)):
        got = BM25Okapi(corpus, k1=k1, b=b)
        ref = osp.BM25OkapiRef(corpus, k1=k1, b=b)
        assert list(got.idf.items()) == list(ref.idf.items())
        assert got.avgdl == ref.avgdl and got.doc_len == ref.doc_len
        for q in (["w0", "w3", "w3", "zzz"], ["w399"], ["w1"] * 5):
            assert np.array_equal(got.get_scores(q), ref.get_scores(q))
    indptr, w = _corpus(rng, 4000, 900, 25)
    got = BM25Okapi.from_term_ids(indptr, w, k1=1.0, b=0.65)
    ref = osp.BM25OkapiRef([w[indptr[i]:indptr[i + 1]].tolist() for i in range(4000)], k1=1.0, b=0.65)
    assert list(got.idf.items()) == list(ref.idf.items())
    q = [int(w[0]), int(w[5]), 899, 12345]
    assert np.array_equal(got.get_scores(q), ref.get_scores(q))
