"""The slice of the faiss Python API that RAG-system uses, served by the MI355X engine.

Reference call sites (vector_storage.py): faiss.IndexFlatIP(d) :79,:146,:151; index.add(x)
:105,:152; index.search(x, k) :196; index.ntotal :193; faiss.write_index :231;
faiss.read_index :298.  IndexFlatL2 is named by BASELINE.json's north_star.

`import rag_system_amd.faiss_compat as faiss` is the whole integration.

The numeric work happens in librag_mi355x.so (HIP, gfx950).  There is no CPU fallback:
constructing an index without the library or without a GPU raises EngineUnavailable.
`set_backend_factory` exists so that host-logic tests can inject the CPU oracle; nothing in
the product calls it.
"""
import contextlib
import ctypes as C
import struct
import threading

import numpy as np

from . import _lib

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


class _HipFlatBackend:
    """rag_flat_t handle (include/rag_mi355x.h)."""

    def __init__(self, d, metric, storage="f32"):
        if storage not in ("f32", "bf16"):
            raise ValueError("storage must be 'f32' or 'bf16', got %r" % (storage,))
        self._lib = _lib.engine()
        h = C.c_void_p()
        dtype = _lib.DTYPE_BF16 if storage == "bf16" else _lib.DTYPE_F32
        _lib.check(self._lib.rag_flat_create(int(d), int(metric), dtype, C.byref(h)), self._lib)
        self._h = h
        self.d = int(d)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.rag_flat_free(h)
            except Exception:
                pass

    @property
    def ntotal(self):
        return int(self._lib.rag_flat_ntotal(self._h))

    def add(self, x):
        _lib.check(self._lib.rag_flat_add(self._h, x.shape[0], _lib.ptr(x)), self._lib)

    def add_synthetic(self, n, seed, first_global_row):
        _lib.check(self._lib.rag_flat_add_synthetic(self._h, int(n), int(seed), int(first_global_row)), self._lib)

    def reserve(self, n):
        _lib.check(self._lib.rag_flat_reserve(self._h, int(n)), self._lib)

    def search(self, q, k):
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(self._lib.rag_flat_search(self._h, nq, _lib.ptr(q), int(k), _lib.ptr(D), _lib.ptr(I)), self._lib)
        return D, I

    def search_begin(self, q, k):
        """First half of search(): enqueues the search on the handle's own stream and returns; the
        result is delivered by search_end().  One begun search per index."""
        _lib.check(self._lib.rag_flat_search_begin(self._h, q.shape[0], _lib.ptr(q), int(k)), self._lib)
        return (q.shape[0], int(k))

    def search_end(self, pending):
        nq, k = pending
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(self._lib.rag_flat_search_end(self._h, _lib.ptr(D), _lib.ptr(I)), self._lib)
        return D, I

    def search_dev(self, nq, q_dev, k, D_dev, I_dev, id_base=0, stream=0):
        """Asynchronous on `stream` (a hipStream_t as int, e.g. torch.cuda.current_stream().cuda_stream;
        0 = HIP's null stream = torch's default stream): ordered with the caller's other work there."""
        _lib.check(self._lib.rag_flat_search_dev(self._h, int(nq), C.c_void_p(q_dev), int(k), C.c_void_p(D_dev),
                                                 C.c_void_p(I_dev), int(id_base), C.c_void_p(int(stream or 0))),
                   self._lib)

    def reset(self):
        _lib.check(self._lib.rag_flat_reset(self._h), self._lib)

    def prefilter(self, enable):
        _lib.check(self._lib.rag_flat_prefilter(self._h, 1 if enable else 0), self._lib)

    def prefilter_redone(self):
        n = C.c_int64()
        _lib.check(self._lib.rag_flat_prefilter_redone(self._h, C.byref(n)), self._lib)
        return n.value

    def last_kernel(self):
        """Which kernels served the last search on this handle (rag_flat_last_kernel)."""
        return (self._lib.rag_flat_last_kernel(self._h) or b"").decode()

    def to_host(self, first=0, n=None):
        n = self.ntotal - first if n is None else n
        out = np.empty((n, self.d), dtype=np.float32)
        _lib.check(self._lib.rag_flat_copy_to_host(self._h, int(first), int(n), _lib.ptr(out)), self._lib)
        return out

    def profiling(self, enable):
        _lib.check(self._lib.rag_flat_profiling(self._h, 1 if enable else 0), self._lib)

    def timing(self):
        n = C.c_int64()
        ms = C.c_double()
        b = C.c_double()
        _lib.check(self._lib.rag_flat_timing(self._h, C.byref(n), C.byref(ms), C.byref(b)), self._lib)
        return n.value, ms.value, b.value


_backend_factory = _HipFlatBackend


def set_backend_factory(factory):
    """Test seam: factory(d, metric) -> backend.  None restores the HIP engine."""
    global _backend_factory
    _backend_factory = factory or _HipFlatBackend


class IndexFlat:
    """faiss.IndexFlat: exact search over raw float32 vectors.

    storage="bf16" (no counterpart in the reference; BASELINE.json config 3): rows are kept
    as bfloat16 (round-to-nearest-even) and queries are rounded the same way at search time;
    scores are the fp32-accumulated products of the rounded operands.  reconstruct_n /
    write_index then return the rounded rows."""

    def __init__(self, d, metric=METRIC_INNER_PRODUCT, storage="f32"):
        if int(d) <= 0:
            raise ValueError("d must be positive")
        self.d = int(d)
        self.metric_type = int(metric)
        self.is_trained = True
        self.storage = storage
        self._b = (_backend_factory(self.d, self.metric_type) if storage == "f32"
                   else _backend_factory(self.d, self.metric_type, storage))
        # A handle takes ONE begun search at a time and refuses every other search until it is ended
        # (rag_flat_search_begin).  The reference's UI starts one thread per query (UI_manager.py:883), so the
        # searches of different threads are serialised here: search_begin holds the lock until search_end, a
        # blocking search waits for it.  (A thread that begins twice without ending is a caller's bug: it is
        # not made to wait for itself, the library's refusal reaches it.)
        self._search_lock = threading.Lock()
        self._begun_by = None

    @property
    def ntotal(self):
        return self._b.ntotal

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("add expects a (n, %d) float32 array, got %s" % (self.d, x.shape))
        if x.shape[0]:
            self._b.add(x)

    def search(self, x, k):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("search expects a (nq, %d) float32 array, got %s" % (self.d, x.shape))
        k = int(k)
        if k <= 0:
            raise ValueError("k must be positive")
        if x.shape[0] == 0:
            return np.empty((0, k), dtype=np.float32), np.empty((0, k), dtype=np.int64)
        with self._serialised():
            return self._b.search(x, k)

    @contextlib.contextmanager
    def _serialised(self):
        if self._begun_by == threading.get_ident():
            yield       # this thread's own begun search is pending: let the library say so
            return
        with self._search_lock:
            yield

    def search_begin(self, x, k):
        """search() in two halves (not in faiss): returns a token for search_end().  The dense pass runs
        on the index's own stream while the caller does something else -- rag_system_amd.search_engine
        computes the BM25 ranking of the same query meanwhile (search_engine.py:266-267)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("search expects a (nq, %d) float32 array, got %s" % (self.d, x.shape))
        k = int(k)
        if k <= 0:
            raise ValueError("k must be positive")
        small = x.shape[0] * (self.d * 4 + 12 * k) + 64 <= (4 << 20)   # fits the pinned staging buffer
        if x.shape[0] == 0 or not small or not hasattr(self._b, "search_begin"):
            return ("done", self.search(x, k))        # nothing to overlap (or a backend without the split)
        if self._begun_by == threading.get_ident():
            return ("pending-unlocked", self._b.search_begin(x, k))    # refused by the library with its message
        self._search_lock.acquire()
        try:
            tok = self._b.search_begin(x, k)
        except BaseException:
            self._search_lock.release()
            raise
        self._begun_by = threading.get_ident()
        return ("pending", tok)

    def search_end(self, token):
        kind, val = token
        if kind == "done":
            return val
        try:
            return self._b.search_end(val)
        finally:
            if kind == "pending":
                self._begun_by = None
                self._search_lock.release()

    def reset(self):
        self._b.reset()

    def set_prefilter(self, enable=True):
        """Not in faiss, off by default (rag_flat_prefilter): keep a bfloat16 copy of the fp32 rows (+50 % HBM);
        searches with k <= 26 read the copy for candidates, score them again from the fp32 rows and verify
        per query that no other row can reach the top k -- the result is that of the fp32 rows."""
        if hasattr(self._b, "prefilter"):      # (test backends that are exact anyway have none)
            self._b.prefilter(enable)

    def reconstruct_n(self, n0=0, ni=None):
        return self._b.to_host(n0, ni)


class IndexFlatIP(IndexFlat):
    def __init__(self, d):
        super().__init__(d, METRIC_INNER_PRODUCT)


class IndexFlatL2(IndexFlat):
    def __init__(self, d):
        super().__init__(d, METRIC_L2)


# ---- index.faiss codec (flat indexes) ----------------------------------------------------
# faiss index_write.cpp / index_read.cpp, little-endian, no padding [published format;
# un-verified against a faiss-written file in this environment -- SURVEY 8(b) b3]:
#   char[4] fourcc "IxFI" (IP) | "IxF2" (L2);  int32 d;  int64 ntotal;  int64 dummy = 1<<20;
#   int64 dummy = 1<<20;  uint8 is_trained;  int32 metric_type (0 IP, 1 L2);
#   uint64 count = ntotal*d;  count x float32 (row-major).
_HDR = struct.Struct("<4siqqqBi")  # 4+4+8+8+8+1+4 = 37 bytes, then uint64 size
_FOURCC = {METRIC_INNER_PRODUCT: b"IxFI", METRIC_L2: b"IxF2"}


def write_index(index, path):
    if not isinstance(index, IndexFlat):
        raise TypeError("write_index supports flat indexes only")
    n = index.ntotal
    with open(path, "wb") as f:
        f.write(_HDR.pack(_FOURCC[index.metric_type], index.d, n, 1 << 20, 1 << 20, 1, index.metric_type))
        f.write(struct.pack("<Q", n * index.d))
        step = max(1, (64 << 20) // (4 * index.d))
        for first in range(0, n, step):
            cnt = min(step, n - first)
            f.write(index.reconstruct_n(first, cnt).astype("<f4", copy=False).tobytes())


def read_index(path, first=0, count=None):
    """first / count (not in faiss): read only rows [first, first + count) of the file -- one rank's block of a
    row-sharded store (rag_system_amd.sharded), so that a 300 GB matrix is never held by one GPU."""
    with open(path, "rb") as f:
        hdr = f.read(_HDR.size)
        if len(hdr) != _HDR.size:
            raise ValueError("%s: truncated faiss index header" % path)
        fourcc, d, ntotal, _d1, _d2, _trained, metric = _HDR.unpack(hdr)
        if fourcc not in (b"IxFI", b"IxF2", b"IxFl"):
            raise ValueError("%s: index type %r is not a flat index" % (path, fourcc))
        if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
            raise ValueError("%s: unsupported metric_type %d" % (path, metric))
        if fourcc == b"IxFI" and metric != METRIC_INNER_PRODUCT or fourcc == b"IxF2" and metric != METRIC_L2:
            raise ValueError("%s: fourcc %r contradicts metric_type %d" % (path, fourcc, metric))
        (n_values,) = struct.unpack("<Q", f.read(8))
        if n_values != ntotal * d:
            raise ValueError("%s: vector payload size %d != ntotal*d = %d" % (path, n_values, ntotal * d))
        index = IndexFlat(d, metric)
        step = max(1, (64 << 20) // (4 * d))
        first = max(0, min(int(first), ntotal))
        left = ntotal - first if count is None else max(0, min(int(count), ntotal - first))
        index.file_ntotal = ntotal          # rows in the file (= ntotal unless a block was asked for)
        if first:
            f.seek(first * d * 4, 1)
        while left > 0:
            cnt = min(step, left)
            buf = f.read(cnt * d * 4)
            if len(buf) != cnt * d * 4:
                raise ValueError("%s: truncated vector payload" % path)
            index.add(np.frombuffer(buf, dtype="<f4").reshape(cnt, d))
            left -= cnt
    return index
