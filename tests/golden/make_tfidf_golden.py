"""Generates tests/golden/tfidf_golden.npz with the scikit-learn installed in the build
container (the library the reference calls at vector_storage.py:216-217 and
search_engine.py:284-285).  Run from the repo root:  python tests/golden/make_tfidf_golden.py

Stored: the fitted CSR matrix, the transformed (un-normalised-by-cosine) query rows, and
cosine_similarity(query, matrix)[0] for each query -- inputs and expected outputs only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from sklearn.feature_extraction.text import TfidfVectorizer  # noqa: E402
from sklearn.metrics.pairwise import cosine_similarity  # noqa: E402
import sklearn  # noqa: E402

from _data import zipf_corpus, zipf_queries  # noqa: E402


def main():
    docs = [" ".join(t) for t in zipf_corpus(400, 1500, 60, seed=11)]
    docs[17] = docs[3]                      # exact duplicate document -> exact score ties
    docs[250] = "zzzunique onlyhere " * 3   # a document sharing no term with most queries
    queries = [" ".join(t) for t in zipf_queries(12, 1500, seed=12)]
    queries.append("zzzunique")             # hits exactly one document
    queries.append("notinvocabularyatall")  # empty query vector -> all zeros
    queries.append(docs[3])                 # a whole document as query
    vec = TfidfVectorizer()
    M = vec.fit_transform(docs)
    out = {
        "sklearn_version": np.array(sklearn.__version__),
        "indptr": M.indptr.astype(np.int64), "indices": M.indices.astype(np.int32), "data": M.data,
        "shape": np.array(M.shape, dtype=np.int64),
        "n_queries": np.array(len(queries)),
    }
    for i, q in enumerate(queries):
        qv = vec.transform([q])
        qv.sort_indices()
        out["q%d_indices" % i] = qv.indices.astype(np.int32)
        out["q%d_data" % i] = qv.data
        out["q%d_scores" % i] = cosine_similarity(qv, M)[0]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tfidf_golden.npz"), **out)
    print("wrote tfidf_golden.npz:", M.shape, "nnz", M.nnz, "queries", len(queries))


if __name__ == "__main__":
    main()
