"""rag_system_amd -- MI355X-native engine behind RAG-system's search path.

Drop-in layers (same names and semantics as the reference's rag_system package):
  faiss_compat   IndexFlatIP / IndexFlatL2 / read_index / write_index  (vector_storage.py:79,196,231,298)
  bm25           BM25Okapi                                              (vector_storage.py:170,178,305)
  tfidf          TF-IDF cosine scorer                                   (search_engine.py:284-285,330-331)
  vector_storage VectorStorage, VectorStorageManager                    (vector_storage.py:58,314)
  search_engine  SearchEngine, detect_knee                              (search_engine.py:37,51)

All arithmetic runs in librag_mi355x.so (hand-written HIP for gfx950, C ABI in
include/rag_mi355x.h).  Importing the package does not touch the GPU; the first numeric call
does, and raises EngineUnavailable when the library or the GPU is missing (no CPU fallback).
"""
__version__ = "0.1"
