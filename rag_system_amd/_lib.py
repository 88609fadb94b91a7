"""ctypes binding of librag_mi355x.so (include/rag_mi355x.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
visible, every numeric entry point raises `EngineUnavailable`.  Build the library with
`python -m rag_system_amd.build` (cross-compiles on a CPU-only machine).
"""
import ctypes as C
import os
import threading

import numpy as np

LIB_NAME = "librag_mi355x.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

RAG_OK = 0
RAG_ERR_INVALID_ARG = -1
RAG_ERR_NO_DEVICE = -2
RAG_ERR_NOT_INIT = -3
RAG_ERR_HIP = -4
RAG_ERR_OOM = -5
RAG_ERR_UNSUPPORTED = -6

METRIC_IP = 0
METRIC_L2 = 1
DTYPE_F32 = 0
DTYPE_BF16 = 1
MAX_K = 256


class EngineError(RuntimeError):
    """A librag_mi355x call failed (message = rag_last_error())."""

    def __init__(self, code, msg):
        super().__init__("librag_mi355x error %d: %s" % (code, msg))
        self.code = code


class EngineUnavailable(EngineError):
    """The HIP library is not built, or no MI355X is visible.  Never falls back to CPU."""


_p = C.c_void_p
_i64 = C.c_int64
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); must list every symbol include/rag_mi355x.h declares.
SIGNATURES = {
    "rag_init": (C.c_int, [C.c_int]),
    "rag_last_error": (C.c_char_p, []),
    "rag_device_count": (C.c_int, []),
    "rag_abi_version": (C.c_int, []),
    "rag_lib_stream": (C.c_int, [C.POINTER(_p)]),
    "rag_flat_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_p)]),
    "rag_flat_free": (None, [_p]),
    "rag_flat_reserve": (C.c_int, [_p, _i64]),
    "rag_flat_add": (C.c_int, [_p, _i64, _p]),
    "rag_flat_add_dev": (C.c_int, [_p, _i64, _p, _p]),
    "rag_flat_add_synthetic": (C.c_int, [_p, _i64, C.c_uint64, _i64]),
    "rag_flat_ntotal": (_i64, [_p]),
    "rag_flat_dim": (C.c_int, [_p]),
    "rag_flat_last_kernel": (C.c_char_p, [_p]),
    "rag_flat_reset": (C.c_int, [_p]),
    "rag_flat_search": (C.c_int, [_p, _i64, _p, C.c_int, _p, _p]),
    "rag_flat_prefilter": (C.c_int, [_p, C.c_int]),
    "rag_flat_prefilter_redone": (C.c_int, [_p, _p]),
    "rag_flat_search_begin": (C.c_int, [_p, _i64, _p, C.c_int]),
    "rag_flat_search_end": (C.c_int, [_p, _p, _p]),
    "rag_flat_search_dev": (C.c_int, [_p, _i64, _p, C.c_int, _p, _p, _i64, _p]),
    "rag_flat_search_dev_noalloc": (C.c_int, [_p, _i64, _p, C.c_int, _p, _p, _i64, _p]),
    "rag_flat_copy_to_host": (C.c_int, [_p, _i64, _i64, _p]),
    "rag_topk_merge_dev": (C.c_int, [C.c_int, C.c_int, _i64, C.c_int, _p, _p, _p, _p, _p]),
    "rag_topk_merge_strided_dev": (C.c_int, [C.c_int, C.c_int, _i64, C.c_int, _p, _p, _i64, _i64, _p, _p, _p]),
    "rag_flat_profiling": (C.c_int, [_p, C.c_int]),
    "rag_flat_timing": (C.c_int, [_p, _i64p, _f64p, _f64p]),
    "rag_set_option": (C.c_int, [C.c_char_p, _i64]),
    "rag_csc_create": (C.c_int, [_i64, _i64, _p, _p, _p, C.POINTER(_p)]),
    "rag_postings_build": (C.c_int, [_i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "rag_csc_free": (None, [_p]),
    "rag_bm25_set": (C.c_int, [_p, _p, _p, C.c_double, C.c_double, C.c_double]),
    "rag_bm25_scores": (C.c_int, [_p, C.c_int, _p, _p, C.c_int, _p, _p]),
    "rag_tfidf_scores": (C.c_int, [_p, C.c_int, _p, _p, _p, C.c_int, _p, _p]),
    "rag_csc_ranks": (C.c_int, [_p, C.c_int, _p, _p]),
    "rag_csc_gather": (C.c_int, [_p, C.c_int, _p, _p]),
    "rag_csc_count_ahead": (C.c_int, [_p, C.c_int, _p, _p, _p]),
    "rag_tfidf_query": (C.c_int, [_p, C.c_int, _p, _p, C.c_int, _p, _p, C.c_int, _p, _p, _p]),
    "rag_tfidf_query_begin": (C.c_int, [_p, C.c_int, _p, _p, C.c_int]),
    "rag_tfidf_query_end": (C.c_int, [_p, C.c_int, _p, _p, C.c_int, _p, _p, _p]),
    "rag_rrf": (C.c_int, [C.c_int, _p, _p, C.c_double, _p, _p, _p]),
    "rag_mmr": (C.c_int, [C.c_int, C.c_int, _p, _p, C.c_double, C.c_int, _p, _i64, _p, _p]),
    "rag_flat_mmr": (C.c_int, [_p, C.c_int, _p, _p, C.c_double, C.c_int, _p, _i64, _p, _p]),
    "rag_hybrid_rrf_mmr": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, _p, C.c_double, C.c_int, C.c_double, C.c_int, _p,
                                     C.c_double, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "rag_csc_set_doc_tables": (C.c_int, [_p, _p, _p]),
}

_lock = threading.Lock()
_lib = None
_inited_device = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  A
    process that loads this library first and imports torch later (embedding_manager, the
    multi-GPU bench) would otherwise end up with two HIP runtimes, and device pointers of
    one are meaningless to the other.  If torch is installed, map ITS runtime first so that
    librag_mi355x.so's NEEDED libamdhip64.so.7 binds to it; torch itself is not imported.
    """
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch's runtime is already mapped; the SONAME match does the rest
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library(path=None):
    """dlopen the library and set prototypes.  Does not touch the GPU."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get("RAG_MI355X_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise EngineUnavailable(
                RAG_ERR_NO_DEVICE,
                "%s not found; build it with `python -m rag_system_amd.build` "
                "(there is no CPU fallback)" % p)
        _share_hip_runtime_with_torch()
        lib = C.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def last_error(lib=None):
    lib = lib or load_library()
    s = lib.rag_last_error()
    return s.decode("utf-8", "replace") if s else ""


def check(rc, lib=None):
    if rc == RAG_OK:
        return
    msg = last_error(lib)
    if rc in (RAG_ERR_NO_DEVICE, RAG_ERR_NOT_INIT):
        raise EngineUnavailable(rc, msg)
    if rc == RAG_ERR_INVALID_ARG:
        raise ValueError("librag_mi355x: " + msg)
    if rc == RAG_ERR_OOM:
        raise MemoryError("librag_mi355x: " + msg)
    raise EngineError(rc, msg)


def engine(device=None):
    """Library handle with the GPU initialised (one process per GPU).

    device: HIP ordinal; default = $RAG_MI355X_DEVICE, else $LOCAL_RANK, else 0.
    """
    global _inited_device
    lib = load_library()
    if device is None:
        if _inited_device is not None:      # whoever initialised the process chose the device
            return lib
        device = int(os.environ.get("RAG_MI355X_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        if _inited_device is None:
            check(lib.rag_init(int(device)), lib)
            _inited_device = int(device)
        elif _inited_device != int(device):
            raise ValueError("engine already initialised on device %d" % _inited_device)
    return lib


def lib_stream():
    """hipStream_t (int) of the library's own stream: what the host entry points run on."""
    lib = engine()
    out = C.c_void_p()
    check(lib.rag_lib_stream(C.byref(out)), lib)
    return out.value or 0


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def as_f32(x, cols=None):
    a = np.ascontiguousarray(x, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a
