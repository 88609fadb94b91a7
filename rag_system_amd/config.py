"""Config -- the constants the hot path consumes, with the reference's names.

Reference: rag_system/config.py:10-61.  Only what the search path reads is mirrored:
BM25_PARAM_K1 / BM25_PARAM_B (used at vector_storage.py:170) and MMR_PARAM_LAMBDA; the
store / UI settings are kept as plain attributes so callers that read them keep working.
Unlike the reference (config.py:37-39 sets the BM25 / MMR constants only inside the
file-exists branch of load()), they are class attributes here and always defined.

Engine keys (no counterpart in the reference): ENGINE_DEVICE, ENGINE_SHARD.
"""
import json
import os


class Config:
    BASE_DIR = "persistent_stores"
    EMBEDDING_MODEL_KEY = "MiniLM"
    CHUNK_SIZE = 200
    OVERLAP = 100
    CROSS_ENCODER_MODEL = "cross-encoder/ms-marco-MiniLM-L-6-v2"
    DEFAULT_TOP_K = 20
    DEFAULT_THRESHOLD = 0.5
    MIN_CHUNKS = 3
    DEFAULT_RETRIEVAL_MODE = "chunk"
    UI_LAST_STORE = ""
    BM25_PARAM_K1 = 1.0
    BM25_PARAM_B = 0.65
    MMR_PARAM_LAMBDA = 0.85
    ENGINE_DEVICE = None  # HIP ordinal; None = $RAG_MI355X_DEVICE / $LOCAL_RANK / 0
    # Row sharding over the GPUs of one node, one process per GPU (rag_system_amd.sharded):
    # "auto" = VectorStorageManager.load_storage shards over torch.distributed's ranks when a process group with
    # more than one rank is initialised; True / False force it.
    ENGINE_SHARD = "auto"

    _KEYS = ("BASE_DIR", "EMBEDDING_MODEL_KEY", "CHUNK_SIZE", "OVERLAP", "CROSS_ENCODER_MODEL", "DEFAULT_TOP_K",
             "DEFAULT_THRESHOLD", "UI_LAST_STORE", "DEFAULT_RETRIEVAL_MODE", "MIN_CHUNKS")

    @classmethod
    def load(cls, path=None):
        """Read the reference's rag_system.json layout if `path` exists (missing keys keep defaults)."""
        if path and os.path.exists(path):
            with open(path, "r") as f:
                data = json.load(f)
            for key in cls._KEYS:
                if key in data:
                    setattr(cls, key, data[key])
        return cls

    @classmethod
    def save(cls, path):
        data = {key: getattr(cls, key) for key in cls._KEYS}
        data.update(BM25_PARAM_K1=cls.BM25_PARAM_K1, BM25_PARAM_B=cls.BM25_PARAM_B,
                    MMR_PARAM_LAMBDA=cls.MMR_PARAM_LAMBDA)
        with open(path, "w") as f:
            json.dump(data, f, indent=4)
