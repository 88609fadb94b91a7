"""CPU tests of the C-ABI boundary: the library loads, exports every symbol the header
declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rag_mi355x.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rag_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from rag_system_amd import _lib
    lib = _lib.load_library()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "librag_mi355x.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes prototypes and include/rag_mi355x.h disagree"
    assert lib.rag_abi_version() == 2


def test_no_gpu_means_loud_failure_not_fallback():
    from rag_system_amd import _lib
    lib = _lib.load_library()
    if lib.rag_device_count() > 0:
        pytest.skip("a GPU is visible; this test covers the CPU-only container")
    assert lib.rag_init(0) == _lib.RAG_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.rag_last_error()
    h = C.c_void_p()
    assert lib.rag_flat_create(384, 0, 0, C.byref(h)) == _lib.RAG_ERR_NOT_INIT
    with pytest.raises(_lib.EngineUnavailable):
        _lib.engine()
    from rag_system_amd import faiss_compat as faiss
    with pytest.raises(_lib.EngineUnavailable):
        faiss.IndexFlatIP(384)
    from rag_system_amd.bm25 import BM25Okapi
    with pytest.raises(_lib.EngineUnavailable):
        BM25Okapi([["a", "b"], ["b"]])
    from rag_system_amd.candidates import reciprocal_rank_fusion, mmr_select
    with pytest.raises(_lib.EngineUnavailable):
        reciprocal_rank_fusion([["a"]])
    with pytest.raises(_lib.EngineUnavailable):
        mmr_select(np.ones((2, 4), np.float32), np.ones(4, np.float32), [1, 1])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rag_system_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
