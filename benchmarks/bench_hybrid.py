#!/usr/bin/env python3
"""BASELINE config 5: strategy=rrf on 1M synthetic documents -- BM25Okapi + TF-IDF postings
kernels + dense top-k fused by RRF, then MMR -- timed per stage on one MI355X, checked at
full size against the numpy restatement (oracle/sparse.py, oracle/fusion.py), with the
reference-style CPU path timed beside it on a bounded sample.

    python benchmarks/bench_hybrid.py [--docs 1000000 --vocab 200000 --dim 384 --queries 50]

Synthetic corpus: seed 777, vocabulary V, Zipf(1.1) term draws, document
length lognormal with mean 120 tokens, one chunk per document; queries of 3-8 terms from the
same Zipf; dense rows from the engine's generator (seed 1234), queries seed 4321.
Per query (search_engine.py:320-350 with top_k = 10):
    dense   index.search(q, 9*top_k)  -> first 3*top_k distinct documents     (rag_flat_search)
    bm25    top 3*top_k chunks                                                 (rag_bm25_scores)
    tfidf   scores of all documents; top (top_k + |A|) + ranks of A            (rag_tfidf_query)
    rrf     1/(60+rank) over the three rankings                                (rag_rrf)
    mmr     top_k of the fused top 3*top_k, stored vectors as embeddings       (rag_mmr)
Prints one JSON line.

oracle/ is imported here in the two roles bench.py has for it and no other: as the CHECKER of the
recorded results, after the timed region, and as the CPU baseline leg (--no-cpu skips it).  Nothing
that is timed as the engine's work touches it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def log(msg):
    print("[bench_hybrid %6.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


def make_corpus(n_docs, vocab, seed):
    rng = np.random.default_rng(seed)
    sigma = 0.4
    lens = np.clip(rng.lognormal(np.log(120.0) - 0.5 * sigma * sigma, sigma, n_docs), 8, 1000).astype(np.int64)
    indptr = np.concatenate([np.zeros(1, np.int64), np.cumsum(lens)])
    w = np.arange(1, vocab + 1, dtype=np.float64) ** -1.1
    cdf = np.cumsum(w / w.sum())
    terms = np.empty(int(indptr[-1]), dtype=np.int32)
    step = 16_000_000
    for lo in range(0, len(terms), step):           # chunked: bounded temporaries
        hi = min(len(terms), lo + step)
        terms[lo:hi] = np.searchsorted(cdf, rng.random(hi - lo)).astype(np.int32)
    np.minimum(terms, vocab - 1, out=terms)
    nq_terms = rng.integers(3, 9, size=4096)
    return indptr, terms, cdf, nq_terms, rng


def tfidf_matrix(n_docs, n_terms, indptr, rows, tf):
    """TfidfVectorizer defaults on integer tokens: idf = ln((1+n)/(1+df)) + 1, l2 rows."""
    from scipy.sparse import csc_matrix
    from sklearn.preprocessing import normalize
    df = np.diff(indptr)
    idf = np.log((1.0 + n_docs) / (1.0 + df)) + 1.0
    m = csc_matrix((tf * np.repeat(idf, df), rows, indptr), shape=(n_docs, n_terms)).tocsr()
    return normalize(m, copy=False), idf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--queries", type=int, default=50)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--cpu-sample-docs", type=int, default=100_000)
    ap.add_argument("--cpu-queries", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--profile-loop", action="store_true", help="cProfile the timed query loop (stderr)")
    args = ap.parse_args()
    n, V, d, top_k = args.docs, args.vocab, args.dim, args.top_k

    from rag_system_amd import _lib, faiss_compat as faiss
    from rag_system_amd.bm25 import BM25Okapi
    from rag_system_amd.candidates import mmr_select, reciprocal_rank_fusion
    from rag_system_amd.tfidf import TfidfScorer
    _lib.engine(0)

    log("generating %d documents over a %d-term vocabulary" % (n, V))
    indptr, terms, cdf, nq_terms, rng = make_corpus(n, V, 777)
    log("%d tokens; building BM25 postings (k1=1.0, b=0.65: config.py:37-38)" % len(terms))
    bm = BM25Okapi.from_term_ids(indptr, terms, k1=1.0, b=0.65, keep_postings=True)
    p_indptr, p_rows, p_tf = bm.postings
    n_terms = len(p_indptr) - 1
    nnz = len(p_rows)
    log("%d distinct terms, %d postings; building TF-IDF matrix" % (n_terms, nnz))
    M, tfidf_idf = tfidf_matrix(n, n_terms, p_indptr, p_rows, p_tf)
    ts = TfidfScorer(None, M)
    Mc = M.tocsc()
    Mc.sort_indices()
    log("dense index %d x %d" % (n, d))
    index = faiss.IndexFlatIP(d)
    index._b.add_synthetic(n, 1234, 0)
    qgen = faiss.IndexFlatIP(d)
    qgen._b.add_synthetic(args.queries, 4321, 0)
    Qd = qgen.reconstruct_n(0, args.queries)
    del qgen

    queries = []                                 # raw integer tokens, as a tokenizer would hand them over
    for qi in range(args.queries):
        raw = np.minimum(np.searchsorted(cdf, rng.random(int(nq_terms[qi]))), V - 1)
        queries.append([int(t) for t in raw])
    vocab_map = bm._vocab                        # raw integer token -> term id

    stages = ["dense", "bm25", "tfidf", "rrf", "mmr", "total"]
    tsum = {s: 0.0 for s in stages}
    sparse_bytes = {"bm25": 0.0, "tfidf": 0.0}
    results = []
    df_all = np.diff(p_indptr)

    def one_query(qi, record):
        raw = queries[qi]
        tids = np.asarray([vocab_map[t] for t in raw if t in vocab_map], dtype=np.int64)
        t0 = time.perf_counter()
        D, I = index.search(Qd[qi:qi + 1], 9 * top_k)             # search_faiss(document mode, 3*top_k)
        dense_rank = list(dict.fromkeys(int(i) for i in I[0]))[:3 * top_k]
        t1 = time.perf_counter()
        bI, bD = bm.top_k(raw, 3 * top_k)                         # search_bm25(3*top_k)
        bm_rank = [int(i) for i in bI]
        t2 = time.perf_counter()
        fused = reciprocal_rank_fusion([dense_rank, bm_rank], k=60)
        t3 = time.perf_counter()
        # TF-IDF: every document is ranked (search_engine.py:332-336); only the candidates and
        # the head of the ranking can reach the fused top_k (rag_system_amd/search_engine.py:245)
        uq, cnt = np.unique(tids, return_counts=True)
        cand = list(fused)
        tI, tD, ranks = ts.ranks_from_row(uq, cnt * tfidf_idf[uq], top_k + len(cand), np.asarray(cand, dtype=np.int64))
        t4 = time.perf_counter()
        for doc, rk in zip(cand, ranks):
            fused[doc] = fused[doc] + 1.0 / (60 + int(rk))
        for rank, doc in enumerate(int(i) for i in tI):
            if doc not in fused:
                fused[doc] = 0 + 1.0 / (60 + rank)
        order = sorted(fused.items(), key=lambda kv: kv[1], reverse=True)[:top_k]
        t5 = time.perf_counter()
        ids = [doc for doc, _ in order]
        sel = mmr_select(Xh[ids], Qd[qi], [120] * len(ids), 0.5, top_k, None)   # search_engine.py:131-133
        t6 = time.perf_counter()
        if record:
            tsum["dense"] += t1 - t0
            tsum["bm25"] += t2 - t1
            tsum["tfidf"] += t4 - t3
            tsum["rrf"] += (t3 - t2) + (t5 - t4)
            tsum["mmr"] += t6 - t5
            tsum["total"] += t6 - t0
            # postings: row 4 + val 8 + (bm25: doc_len 4) + score read+write 16; plus clearing and
            # reading the n-row score vector (memset + top-k), plus one more read for the ranks
            sparse_bytes["bm25"] += 32.0 * float(df_all[tids].sum()) + 16.0 * n
            sparse_bytes["tfidf"] += 28.0 * float(df_all[uq].sum()) + 24.0 * n
        return dict(dense=dense_rank, bm25=(bm_rank, bD), tfidf=([int(i) for i in tI], tD), cand=cand,
                    ranks=[int(r) for r in ranks], fused=order, mmr=[ids[i] for i in sel], tids=tids)

    log("copying the dense rows to the host once (MMR candidate embeddings)")
    Xh = index.reconstruct_n(0, n)
    log("warm-up")
    for qi in range(min(3, args.queries)):
        one_query(qi, False)
    log("timing %d queries" % args.queries)
    prof = None
    if args.profile_loop:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    for qi in range(args.queries):
        results.append(one_query(qi, True))
    if prof is not None:
        import io
        import pstats
        prof.disable()
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(30)
        print(buf.getvalue(), file=sys.stderr)
    nq = float(args.queries)
    per_stage_ms = {s: tsum[s] / nq * 1e3 for s in stages}

    # ---- parity at full size against the numpy restatement (oracle/)
    ncheck = min(8, args.queries)
    log("parity check against oracle/ on %d queries" % ncheck)
    from oracle import dense as od
    from oracle import fusion as ofu
    from oracle import sparse as osp
    all_ids = np.arange(n)
    Dr, Ir, D64 = od.flat_search_synth(1234, 0, n, d, Qd[:ncheck], 9 * top_k)
    for qi in range(ncheck):
        r, tids = results[qi], results[qi]["tids"]
        s_ref = osp.bm25_scores_postings(p_indptr, p_rows, p_tf, bm._idf_arr, np.asarray(bm.doc_len), bm.avgdl,
                                         1.0, 0.65, tids)
        top = np.lexsort((all_ids, -s_ref))[:3 * top_k]
        assert top.tolist() == r["bm25"][0], "BM25 ids differ from the restatement (query %d)" % qi
        assert s_ref[top].tobytes() == np.asarray(r["bm25"][1]).tobytes(), "BM25 scores differ bitwise"
        uq, cnt = np.unique(tids, return_counts=True)
        w = cnt * tfidf_idf[uq]
        w = w / np.sqrt((w * w).sum())
        t_ref = osp.tfidf_scores_postings(Mc.indptr, Mc.indices, Mc.data, uq, w, n)
        t_order = np.lexsort((all_ids, -t_ref))
        assert t_order[:len(r["tfidf"][0])].tolist() == r["tfidf"][0], "TF-IDF ids differ (query %d)" % qi
        assert np.allclose(t_ref[t_order[:len(r["tfidf"][0])]], r["tfidf"][1], rtol=1e-12, atol=0), "TF-IDF scores differ"
        full_rank = np.empty(n, dtype=np.int64)
        full_rank[t_order] = all_ids
        assert full_rank[r["cand"]].tolist() == r["ranks"], "TF-IDF ranks differ (query %d)" % qi
        assert list(dict.fromkeys(int(i) for i in Ir[qi]))[:3 * top_k] == r["dense"], "dense ids differ (query %d)" % qi
        # the reference's RRF: every entry of the three FULL rankings (search_engine.py:352-359)
        fused = ofu.reciprocal_rank_fusion([[(i, 0.0) for i in r["dense"]], [(i, 0.0) for i in r["bm25"][0]],
                                            [(int(i), 0.0) for i in t_order]], k=60)
        order = sorted(fused.items(), key=lambda kv: kv[1], reverse=True)[:top_k]
        assert order == r["fused"], "fused top_k differs from RRF over the full rankings (query %d)" % qi
        ids = [a for a, _ in order]
        sel = ofu.mmr_selection(Xh[ids], Qd[qi], [120] * len(ids), 0.5, top_k, None)
        assert [ids[i] for i in sel] == r["mmr"], "MMR order differs (query %d)" % qi
    checked = ncheck

    out = {
        "metric": "QPS, strategy=rrf (dense + BM25Okapi + TF-IDF + RRF + MMR) over %d synthetic documents" % n,
        "value": nq / tsum["total"], "unit": "QPS", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
        "dtype": "f64 (sparse scoring, RRF), f32 (dense)",
        "config": {"workload": "%d docs, vocab %d (%d seen), %d postings, Zipf(1.1), mean length 120; dense %d x %d fp32; "
                               "top_k=%d; %d queries of 3-8 terms" % (n, V, n_terms, nnz, n, d, top_k, args.queries)},
        "ms_per_query": tsum["total"] / nq * 1e3,
        "stage_ms": per_stage_ms,
        "sparse_roofline": {
            s: {"bound": "hbm", "achieved": sparse_bytes[s] / tsum[s] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": sparse_bytes[s] / tsum[s] / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_query": sparse_bytes[s] / nq,
                "note": "wall clock of the whole C-ABI call (launch per query term + top-k + D2H + sync)"}
            for s in ("bm25", "tfidf")},
        "parity": {"queries_checked": checked, "bm25": "ids and float64 scores bit-exact", "tfidf": "ids exact, scores 1e-12",
                   "ranks": "exact", "dense": "ids exact", "rrf": "identical to RRF over the full rankings",
                   "mmr": "identical order"},
    }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_leg(args, indptr, terms, queries, bm, Qd, top_k)
    print(json.dumps(out), flush=True)


def cpu_leg(args, indptr, terms, queries, bm, Qd, top_k):
    """The reference's own per-query work (rank_bm25 dict loop + Python sorts + sklearn cosine
    + faiss-style flat scan) restated on the first `cpu_sample_docs` documents, one core."""
    from oracle import dense as od
    from oracle import sparse as osp
    from sklearn.metrics.pairwise import cosine_similarity
    m = min(args.cpu_sample_docs, args.docs)
    log("CPU leg: restatement on the first %d documents" % m)
    corpus = [terms[indptr[i]:indptr[i + 1]].tolist() for i in range(m)]
    ref = osp.BM25OkapiRef(corpus, k1=1.0, b=0.65)
    M, _ = None, None
    X = od.synth_rows(1234, 0, m, args.dim)
    from scipy.sparse import csr_matrix
    from sklearn.feature_extraction.text import TfidfTransformer
    lens = np.diff(indptr[:m + 1])
    counts = csr_matrix((np.ones(int(lens.sum())), terms[:int(indptr[m])], indptr[:m + 1]),
                        shape=(m, args.vocab))
    counts.sum_duplicates()
    tr = TfidfTransformer()
    Mtf = tr.fit_transform(counts)
    t0 = time.perf_counter()
    nq = min(args.cpu_queries, len(queries))
    for qi in range(nq):
        raw = queries[qi]
        s = ref.get_scores(raw)
        sorted(range(len(s)), key=lambda i: s[i], reverse=True)[:3 * top_k]
        qv = tr.transform(csr_matrix((np.ones(len(raw)), (np.zeros(len(raw), int), raw)), shape=(1, args.vocab)))
        sc = cosine_similarity(qv, Mtf)[0]
        sorted(((i, float(sc[i])) for i in range(m)), key=lambda x: x[1], reverse=True)
        od.cpu_baseline_search(X, Qd[qi:qi + 1], 9 * top_k, 1)
    el = time.perf_counter() - t0
    scale = m / float(args.docs)
    return {"value": nq / el * scale, "unit": "QPS", "cores": 1, "kind": "port",
            "sample": "%d queries over the first %d of %d documents (%.1f s): rank_bm25-style dict loop + Python "
                      "sorts over all documents + sklearn cosine + flat fp32 scan; QPS scaled by the document "
                      "ratio (every term is linear in the document count)" % (nq, m, args.docs, el)}


if __name__ == "__main__":
    T0 = time.perf_counter()
    main()
