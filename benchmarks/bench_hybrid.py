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


from benchmarks.hybrid_workload import HybridWorkload, STAGES  # noqa: E402


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
    ap.add_argument("--check-queries", type=int, default=8, help="queries checked against oracle/ after the timed loop")
    ap.add_argument("--search-engine", action="store_true",
                    help="also time the queries through rag_system_amd.SearchEngine on a VectorStorage over the same indices")
    ap.add_argument("--no-tail", action="store_true", help="the three-call route instead of rag_hybrid_rrf_mmr")
    ap.add_argument("--profile-loop", action="store_true", help="cProfile the timed query loop (stderr)")
    args = ap.parse_args()
    n, V, d, top_k = args.docs, args.vocab, args.dim, args.top_k

    from rag_system_amd import _lib
    _lib.engine(0)
    wl = HybridWorkload(n, V, d, args.queries, top_k, log)
    n_terms, nnz = wl.n_terms, wl.nnz
    stages = STAGES
    tsum = {s: 0.0 for s in stages}
    sparse_bytes = {"bm25": 0.0, "tfidf": 0.0}
    results = []
    log("warm-up")
    for qi in range(min(10, args.queries)):     # first calls size workspaces and set kernel attributes
        wl.query(qi, tail=not args.no_tail)
    log("timing %d queries" % args.queries)
    prof = None
    if args.profile_loop:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    for qi in range(args.queries):
        results.append(wl.query(qi, tsum, sparse_bytes, keep_emb=qi < 8, tail=not args.no_tail))
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
    ncheck = min(args.check_queries, args.queries)
    log("parity check against oracle/ on %d queries" % ncheck)
    checked = wl.check(list(range(ncheck)), results[:ncheck])

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
                "note": "algorithmic bytes / wall clock of the whole C-ABI call (scoring launch + selection launch, results in pinned memory, one synchronise); kernel-time figures: DESIGN.md 4.3"}
            for s in ("bm25", "tfidf")},
        "parity": {"queries_checked": checked, "bm25": "ids and float64 scores bit-exact", "tfidf": "ids exact, scores 1e-12",
                   "ranks": "exact", "dense": "ids exact", "rrf": "identical to RRF over the full rankings",
                   "mmr": "identical order"},
    }
    if args.search_engine:
        from benchmarks.hybrid_workload import make_search_engine, time_search_engine
        eng, _store, qtexts = make_search_engine(wl, log)
        out["through_search_engine"] = time_search_engine(wl, eng, qtexts, args.queries, top_k,
                                                          {qi: results[qi] for qi in range(ncheck)})
    if not args.no_cpu:
        from benchmarks.hybrid_workload import cpu_baseline
        out["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample_docs, args.cpu_queries, log)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    T0 = time.perf_counter()
    main()
