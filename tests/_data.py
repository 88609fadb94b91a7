"""Seeded synthetic text / token corpora shared by the tests (no external data)."""
import numpy as np

_SYL = ["ka", "lo", "mi", "ra", "te", "su", "no", "vi", "pa", "de", "gu", "zo", "be", "chi", "fra", "sto"]


def word(i):
    """Deterministic pronounceable token for vocabulary index i (>= 2 chars, \\w only)."""
    s = ""
    i += 1
    while True:
        s += _SYL[i % len(_SYL)]
        i //= len(_SYL)
        if i == 0:
            break
    return s


def zipf_corpus(n_docs, vocab, mean_len, seed, a=1.1):
    """List of token lists; term ranks ~ Zipf(a) truncated to `vocab`, lengths ~ lognormal."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, vocab + 1) ** a
    p /= p.sum()
    lens = np.maximum(1, rng.lognormal(np.log(mean_len), 0.5, n_docs).astype(np.int64))
    flat = rng.choice(vocab, size=int(lens.sum()), p=p)
    out, o = [], 0
    for L in lens:
        out.append([word(int(t)) for t in flat[o:o + L]])
        o += L
    return out


def zipf_queries(n_q, vocab, seed, lo=3, hi=8, a=1.1):
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, vocab + 1) ** a
    p /= p.sum()
    return [[word(int(t)) for t in rng.choice(vocab, size=int(rng.integers(lo, hi + 1)), p=p)] for _ in range(n_q)]
