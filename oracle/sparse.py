"""CPU restatement of the sparse scoring path -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

BM25: the arithmetic lives in rank_bm25 (unpinned, requirements.txt:12, absent here; its
API and formula are those of rank_bm25 0.2.x).  The reference holds no fixture for it; the only
output of the real library available offline is the example its README publishes, which the class below
reproduces (tests/golden/published_answers.json, tests/test_published_answers.py) -- beyond that example:
PARITY UNPINNED.  The class below restates the published algorithm of
rank_bm25.BM25 / BM25Okapi line by line (pure Python + numpy, float64), as called at
vector_storage.py:170 (BM25Okapi(corpus, k1, b)), :305 (defaults k1=1.5, b=0.75,
epsilon=0.25) and :178 (get_scores).

TF-IDF: scikit-learn IS importable here, so the oracle is the real thing:
TfidfVectorizer().fit_transform (vector_storage.py:216-217), vectorizer.transform and
sklearn.metrics.pairwise.cosine_similarity (search_engine.py:284-285, 330-331).  Golden
fixtures made with it are committed under tests/golden/ (tests/golden/make_tfidf_golden.py).
"""
import math

import numpy as np


class BM25OkapiRef:
    """rank_bm25.BM25Okapi restated (0.2.x)."""

    def __init__(self, corpus, k1=1.5, b=0.75, epsilon=0.25):
        self.k1 = k1
        self.b = b
        self.epsilon = epsilon
        self.corpus_size = 0
        self.avgdl = 0
        self.doc_freqs = []
        self.idf = {}
        self.doc_len = []
        nd = self._initialize(corpus)
        self._calc_idf(nd)

    def _initialize(self, corpus):
        nd = {}
        num_doc = 0
        for document in corpus:
            self.doc_len.append(len(document))
            num_doc += len(document)
            frequencies = {}
            for word in document:
                if word not in frequencies:
                    frequencies[word] = 0
                frequencies[word] += 1
            self.doc_freqs.append(frequencies)
            for word, _freq in frequencies.items():
                try:
                    nd[word] += 1
                except KeyError:
                    nd[word] = 1
            self.corpus_size += 1
        self.avgdl = num_doc / self.corpus_size
        return nd

    def _calc_idf(self, nd):
        idf_sum = 0
        negative_idfs = []
        for word, freq in nd.items():
            idf = math.log(self.corpus_size - freq + 0.5) - math.log(freq + 0.5)
            self.idf[word] = idf
            idf_sum += idf
            if idf < 0:
                negative_idfs.append(word)
        self.average_idf = idf_sum / len(self.idf)
        eps = self.epsilon * self.average_idf
        for word in negative_idfs:
            self.idf[word] = eps

    def get_scores(self, query):
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for q in query:
            q_freq = np.array([(doc.get(q) or 0) for doc in self.doc_freqs])
            score += (self.idf.get(q) or 0) * (q_freq * (self.k1 + 1) /
                                               (q_freq + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl)))
        return score


def stable_topk_desc(scores, k):
    """vector_storage.py:180: sorted(range(n), key=lambda i: scores[i], reverse=True)[:k]."""
    return sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)[:k]


def stable_ranks_desc(scores):
    """rank[i] = position of i in the stable descending sort (search_engine.py:332-336)."""
    order = sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)
    ranks = np.empty(len(scores), dtype=np.int64)
    ranks[order] = np.arange(len(scores))
    return ranks


def tfidf_cosine(vectorizer, matrix, query):
    """search_engine.py:284-285: cosine_similarity(vectorizer.transform([query]), matrix)[0]."""
    from sklearn.metrics.pairwise import cosine_similarity
    return cosine_similarity(vectorizer.transform([query]), matrix)[0]


def bm25_scores_postings(indptr, rows, tf, idf, doc_len, avgdl, k1, b, term_ids):
    """BM25OkapiRef.get_scores evaluated on term-major postings: the same float64 operations in
    the same order for every (document, query term) with tf > 0; documents without the term
    receive idf * (0 * (k1+1) / (0 + ...)) = +0.0 in get_scores, which leaves the sum unchanged.
    tests/test_oracle_cpu.py pins it bit-for-bit to BM25OkapiRef.  Used for full-size checks
    where a million Python dicts would not fit."""
    score = np.zeros(len(doc_len))
    dl = np.asarray(doc_len)
    for t in term_ids:
        lo, hi = int(indptr[t]), int(indptr[t + 1])
        r = rows[lo:hi]
        q_freq = tf[lo:hi]
        score[r] += idf[t] * (q_freq * (k1 + 1) / (q_freq + k1 * (1 - b + b * dl[r] / avgdl)))
    return score


def tfidf_scores_postings(indptr, rows, val, term_ids, weights, n_rows):
    """cosine_similarity(q, M)[0] on term-major postings of the row-normalised matrix: scipy's
    csr_matmat adds w_t * M[doc, t] term by term in ascending term order."""
    score = np.zeros(n_rows)
    for t, w in zip(term_ids, weights):
        lo, hi = int(indptr[t]), int(indptr[t + 1])
        score[rows[lo:hi]] += w * val[lo:hi]
    return score
