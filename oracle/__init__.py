"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (rag_system_amd) never does and has no CPU fallback.

Modules
  dense   -- ctypes binding of oracle/flat_ref.c (IndexFlatIP/L2 restatement, synthetic rows)
  sparse  -- BM25Okapi (rank_bm25) and TF-IDF cosine restatements in numpy / pure Python
  fusion  -- RRF, MMR, multi / hybrid fusion, knee detection restated from search_engine.py

Pinning status (see DESIGN.md "Oracle"):
  dense   PARITY UNPINNED  (faiss-cpu absent, reference holds no fixture for it)
  bm25    PARITY UNPINNED  (rank_bm25 absent, reference holds no fixture for it)
  tfidf   pinned by tests/golden/tfidf_*.npz generated with the importable scikit-learn
  fusion  PARITY UNPINNED  (restated from the reference source; it has no tests)
"""
