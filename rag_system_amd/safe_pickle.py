"""Restricted unpickling of the store files.

The reference keeps a store as four files, three of them pickles (vector_storage.py:232-267), and
loads them with pickle.load (:276-311), which executes whatever the file names.  A store directory
copied from elsewhere -- also one written by the reference -- would run arbitrary code on load, or
already on VectorStorageManager.list_persistent_storages().  The layout has to stay (the files are
interchangeable with the reference), so the files are read through an Unpickler that resolves only
what such a store legitimately contains:

  metadata.pkl, index_bm25.pkl   plain containers, strings and numbers           -> load_plain
  tfidf.pkl                      + numpy arrays, a scipy.sparse matrix and
                                   scikit-learn's TfidfVectorizer / TfidfTransformer -> load_tfidf

Anything else raises pickle.UnpicklingError naming the global.  RAG_TRUST_PICKLES=1 restores plain
pickle.load for stores with custom objects that the caller trusts.
"""
import io
import os
import pickle

_BUILTINS = {"dict", "list", "set", "frozenset", "tuple", "str", "bytes", "bytearray", "int", "float", "bool",
             "complex", "slice", "range"}
_PLAIN = {("builtins", n) for n in _BUILTINS} | {("collections", "OrderedDict")}

_NUMPY_MODULES = ("numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric",
                  "numpy._core.numeric", "numpy.core._multiarray_umath", "numpy._core._multiarray_umath")
_NUMPY_NAMES = {"ndarray", "dtype", "_reconstruct", "scalar", "_frombuffer", "float64", "float32", "float16", "int64",
                "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "intc", "intp", "bool_", "bool",
                "longlong", "ulonglong", "str_", "bytes_", "object_"}
_SCIPY_NAMES = {"csr_matrix", "csc_matrix", "coo_matrix", "csr_array", "csc_array", "coo_array"}
_SKLEARN_NAMES = {"TfidfVectorizer", "TfidfTransformer", "CountVectorizer"}


def _tfidf_allowed(module, name):
    if (module, name) in _PLAIN:
        return True
    if module in _NUMPY_MODULES and name in _NUMPY_NAMES:
        return True
    if (module == "scipy.sparse" or module.startswith("scipy.sparse.")) and name in _SCIPY_NAMES:
        return True
    if module in ("sklearn.feature_extraction.text", "sklearn.feature_extraction._tfidf") and name in _SKLEARN_NAMES:
        return True
    return False


class _Restricted(pickle.Unpickler):
    def __init__(self, f, allowed, what):
        super().__init__(f)
        self._allowed = allowed
        self._what = what

    def find_class(self, module, name):
        if self._allowed(module, name):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(
            "%s names %s.%s, which a RAG-system store does not contain; refusing to load it "
            "(RAG_TRUST_PICKLES=1 loads files you trust with plain pickle)" % (self._what, module, name))

    def persistent_load(self, pid):
        raise pickle.UnpicklingError("%s uses persistent ids; refusing to load it" % self._what)


def _load(f, allowed, what):
    if os.environ.get("RAG_TRUST_PICKLES") == "1":
        return pickle.load(f)
    if isinstance(f, (bytes, bytearray)):
        f = io.BytesIO(f)
    return _Restricted(f, allowed, what).load()


def load_plain(f, what="store file"):
    """metadata.pkl / index_bm25.pkl: containers, strings, numbers only."""
    return _load(f, lambda m, n: (m, n) in _PLAIN, what)


def load_tfidf(f, what="tfidf.pkl"):
    """tfidf.pkl: + numpy arrays, scipy.sparse matrices, scikit-learn's TF-IDF vectorizer."""
    return _load(f, _tfidf_allowed, what)
