"""Host-side text helpers around the hot path (not part of it).

word_tokenize: the reference tokenises BM25 chunks and queries with nltk.word_tokenize
(vector_storage.py:162,177).  nltk (and its punkt data, fetched from the network at import in
the reference) is an optional dependency here: when importable it is used, otherwise the Penn
Treebank splitting rules it applies are restated as regular expressions (without punkt's sentence
splitting).  Synthetic
benchmarks use integer term ids, so tokenisation is outside every measured path.

HashingEmbeddingManager: a deterministic, model-free stand-in for the reference's
EmbeddingManager (embedding_manager.py:21) so that the store / search plumbing can run
offline (BASELINE config #1 needs SentenceTransformer weights that cannot be fetched here).
Producing embeddings is outside the hot path; real deployments pass the reference's own
EmbeddingManager (any object with .model.encode, .embed_text, .embed_query works).
"""
import hashlib
import re

import numpy as np

import warnings

NLTK = "nltk.word_tokenize"
FALLBACK = "treebank-regex-fallback"

# The Penn Treebank word-splitting rules nltk.word_tokenize applies to each sentence (nltk.tokenize.treebank,
# restated from its published rule list): quotes to `` / '', commas and colons split unless inside numbers, the
# period that ends the text, ; @ # $ % & ? ! brackets and -- set apart, contractions ('s 'm 'd 'll 're 've n't)
# split off.  What is NOT reproduced is the punkt sentence splitter that runs before it (absent with nltk): a
# period that ends a sentence in the MIDDLE of a text stays attached to its word here ("york." rather than
# "york", "."), as it does in nltk's own TreebankWordTokenizer.  Reproduces the examples nltk documents
# (tests/test_oracle_cpu.py); BM25 text parity with stores tokenised by the real nltk stays "parity unpinned".
_TB_RULES = [(re.compile(p), r) for p, r in (
    (r'^\"', r"``"), (r"(``)", r" \1 "), (r"([ \(\[{<])(\"|\'{2})", r"\1 `` "),
    (r"([:,])([^\d])", r" \1 \2"), (r"([:,])$", r" \1 "), (r"\.\.\.", r" ... "), (r"[;@#$%&]", r" \g<0> "),
    (r"([^\.])(\.)([\]\)}>\"\']*)\s*$", r"\1 \2\3 "), (r"[?!]", r" \g<0> "), (r"([^'])' ", r"\1 ' "),
    (r"[\]\[\(\)\{\}\<\>]", r" \g<0> "), (r"--", r" -- "))]
_TB_RULES2 = [(re.compile(p), r) for p, r in (
    (r'"', " '' "), (r"(\S)(\'\')", r"\1 \2 "), (r"([^' ])('[sS]|'[mM]|'[dD]|') ", r"\1 \2 "),
    (r"([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) ", r"\1 \2 "))]


def _treebank_tokenize(text):
    for rx, rep in _TB_RULES:
        text = rx.sub(rep, text)
    text = " " + text + " "
    for rx, rep in _TB_RULES2:
        text = rx.sub(rep, text)
    return text.split()
_tokenizer = None   # (name, callable), resolved once


def _resolve():
    """nltk.word_tokenize (punkt + treebank: what the reference uses, vector_storage.py:162,177) when nltk
    and its punkt data are usable, else the regex fallback -- decided ONCE per process, with one warning:
    the two agree on contractions, decimals, currency, quotes and brackets (the Treebank rules) and differ where
    punkt would have split sentences (a period ending a sentence in mid-text stays on its word here), so BM25
    scores of a store tokenised by one and queried with the other can differ from the reference's.
    BM25 TEXT parity is therefore 'parity unpinned' without nltk; the numeric parity tests use token lists."""
    global _tokenizer
    if _tokenizer is None:
        try:
            from nltk.tokenize import word_tokenize as _nltk_tok
            _nltk_tok("probe the punkt data, don't skip")
            _tokenizer = (NLTK, _nltk_tok)
        except Exception:  # nltk or its punkt data missing
            warnings.warn("nltk (or its punkt data) is not available: BM25 text is tokenised by a regex fallback; "
                          "scores of stores tokenised by the reference's nltk.word_tokenize are not guaranteed "
                          "to match", RuntimeWarning, stacklevel=3)
            _tokenizer = (FALLBACK, _treebank_tokenize)
    return _tokenizer


def tokenizer_name():
    return _resolve()[0]


def word_tokenize(text):
    return _resolve()[1](text)


def check_store_tokenizer(stored_name):
    """stored_name: metadata.pkl's "bm25_tokenizer" (None: a store the reference wrote -> nltk)."""
    stored = stored_name or NLTK
    if stored != tokenizer_name():
        warnings.warn("this store's BM25 corpus was tokenised by %s but queries will be tokenised by %s: BM25 "
                      "rankings can differ from the ones the store was built for (rebuild_bm25_index() after "
                      "clearing bm25_corpus re-tokenises)" % (stored, tokenizer_name()), RuntimeWarning, stacklevel=3)


class _HashingModel:
    """encode(list[str]) -> (n, d) float32: signed hashed bag of words (not normalised)."""

    def __init__(self, dim):
        self.dim = dim

    def encode(self, texts):
        if isinstance(texts, str):
            texts = [texts]
        out = np.zeros((len(texts), self.dim), dtype=np.float32)
        for i, t in enumerate(texts):
            for w in re.findall(r"\w+", t.lower()):
                h = int.from_bytes(hashlib.md5(w.encode("utf-8")).digest()[:8], "little")
                out[i, h % self.dim] += 1.0 if (h >> 40) & 1 else -1.0
                out[i, (h >> 20) % self.dim] += 0.5
        return out


class HashingEmbeddingManager:
    """Same surface as the reference's EmbeddingManager: model.encode, embed_text, embed_query,
    chunk_size / overlap / model_name."""

    def __init__(self, dim=384, chunk_size=100, overlap=20, model_name="hashing-stub"):
        self.model = _HashingModel(dim)
        self.model_name = model_name
        self.chunk_size = chunk_size
        self.overlap = overlap
        self.dim = dim

    def set_chunk_params(self, chunk_size, overlap):
        self.chunk_size = chunk_size
        self.overlap = overlap

    def _chunks(self, text):
        words = text.split()
        if len(words) < 50:
            return [text]
        step = max(1, self.chunk_size - self.overlap)
        return [" ".join(words[i:i + self.chunk_size]) for i in range(0, len(words), step)
                if i == 0 or i + self.overlap < len(words)]

    def embed_text(self, text, doc_name=None):
        chunk_texts = self._chunks(text)
        emb = np.array(self.model.encode(chunk_texts), dtype="float32")
        norms = np.linalg.norm(emb, axis=1, keepdims=True)
        norms[norms == 0] = 1e-9
        emb = emb / norms
        chunks = [{"chunk": c, "source": doc_name} for c in chunk_texts] if doc_name is not None else chunk_texts
        return chunks, emb

    def embed_query(self, query):
        vec = self.model.encode([query])[0].astype("float32")
        norm = np.linalg.norm(vec)
        if norm < 1e-9:
            norm = 1.0
        return vec / norm
