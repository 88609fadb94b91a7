"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (rag_system_amd) never does and has no CPU fallback.

Modules
  dense   -- ctypes binding of oracle/flat_ref.c (IndexFlatIP/L2 restatement, synthetic rows)
  sparse  -- BM25Okapi (rank_bm25) and TF-IDF cosine restatements in numpy / pure Python
  fusion  -- RRF, MMR, multi / hybrid fusion, knee detection restated from search_engine.py

Pinning status (see DESIGN.md "Oracle"):
  dense   faiss-cpu absent, the reference holds no fixture for it.  Pinned (round 3) as far as the library's own
          PUBLISHED outputs go: the faiss wiki's "Getting started" IndexFlatL2 example (60 ids, 20 distances;
          tests/golden/published_answers.json, tests/test_published_answers.py).  Beyond that example: parity unpinned.
  bm25    rank_bm25 absent, the reference holds no fixture for it.  Pinned (round 3) by the rank_bm25 README's published
          example (same fixture) and a hand-computed case; beyond that: parity unpinned.
  tfidf   pinned by tests/golden/tfidf_*.npz generated with the importable scikit-learn
  fusion  pinned (round 3) to the outputs of the reference's own rag_system/search_engine.py
          (tests/golden/strategy_golden.json, tests/golden/make_strategy_golden.py, tests/test_strategy_golden.py)
"""
