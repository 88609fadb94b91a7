"""world_size-2 / -4 (gloo, CPU) tests of the multi-GPU path's host logic: shard bounds, global ids,
all-gather + merge == searching the whole corpus; plus the 1/2/4/8-way "fake shard" check of
SURVEY section 4 tier 3 in one process.  The local search and the merge are the CPU
oracle here (injected); on the GPU the same ShardedFlatSearcher drives librag_mi355x.so
(tests/test_flat_gpu.py::test_sharded_search_and_device_merge covers the kernels)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import dense as od
        from rag_system_amd.sharded import ShardedFlatSearcher, shard_bounds
        n, d, nq, k = 5001, 64, 3, 20
        X = od.synth_rows(9, 0, n, d)
        X[4000] = X[10]                        # tie across the shard border
        Q = od.synth_queries(8, nq, d)
        for metric in (0, 1):
            lo, cnt = shard_bounds(n, world)[rank]
            Xl = X[lo:lo + cnt]

            def local_search(q, kk, D_loc, I_loc, _stream):
                D, I, _ = od.flat_search(Xl, q.numpy(), kk, metric)
                D_loc.copy_(torch.from_numpy(D))
                I_loc.copy_(torch.from_numpy(np.where(I >= 0, I + lo, -1)))

            def merge(m, Da, Ia, Do, Io, _stream):
                D, I = od.topk_merge(m, Da.numpy(), Ia.numpy())
                Do.copy_(torch.from_numpy(D))
                Io.copy_(torch.from_numpy(I))

            s = ShardedFlatSearcher(local_search, metric, nq, k, torch.device("cpu"), merge=merge)
            D, I = s.search(torch.from_numpy(Q))
            Dw, Iw, _ = od.flat_search(X, Q, k, metric)
            assert np.array_equal(I.numpy(), Iw), (rank, metric)
            assert np.array_equal(D.numpy(), Dw), (rank, metric)
            # caller-provided output buffer (the bench hands in pinned memory): same packed layout
            out = torch.zeros(ShardedFlatSearcher.words(nq, k), dtype=torch.int64)
            s2 = ShardedFlatSearcher(local_search, metric, nq, k, torch.device("cpu"), merge=merge, out=out)
            s2.search(torch.from_numpy(Q))
            D2, I2 = s2.unpack(out)
            assert np.array_equal(I2.numpy(), Iw) and np.array_equal(D2.numpy(), Dw), (rank, metric)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    from rag_system_amd.sharded import shard_bounds
    assert shard_bounds(10, 1) == [(0, 10)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 1), (2, 0), (2, 0)]
    assert sum(c for _, c in shard_bounds(100_000_000, 8)) == 100_000_000


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_sharded_search_equals_whole(world):
    port = 29500 + (os.getpid() % 2000) + world
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


@pytest.mark.parametrize("metric", [0, 1])
def test_fake_shards_1_2_4_8_way_identical(metric):
    """SURVEY section 4 tier 3: the corpus cut 1/2/4/8 ways (shard_bounds), each shard searched
    alone with global ids, lists merged (score best-first, lower id first) == the whole corpus.
    Includes a shard count that leaves empty shards and ties that cross every border."""
    from oracle import dense as od
    from rag_system_amd.sharded import shard_bounds
    n, d, nq, k = 1003, 48, 4, 25
    X = od.synth_rows(77, 0, n, d)
    for b in (125, 126, 250, 251, 501, 502, 752, 1002):     # duplicates of row 3 on both sides of borders
        X[b] = X[3]
    Q = od.synth_queries(78, nq, d)
    Q[0] = X[3]
    Dw, Iw, _ = od.flat_search(X, Q, k, metric)
    for world in (1, 2, 4, 8):
        Dl, Il = [], []
        for lo, cnt in shard_bounds(n, world):
            D, I, _ = od.flat_search(X[lo:lo + cnt], Q, k, metric)
            Dl.append(D)
            Il.append(np.where(I >= 0, I + lo, -1))
        Dm, Im = od.topk_merge(metric, np.stack(Dl), np.stack(Il))
        assert np.array_equal(Im, Iw) and np.array_equal(Dm, Dw), world
    # more shards than rows
    Xs = X[:3]
    Dw, Iw, _ = od.flat_search(Xs, Q, k, metric)
    Dl, Il = [], []
    for lo, cnt in shard_bounds(3, 8):
        D, I, _ = od.flat_search(Xs[lo:lo + cnt], Q, k, metric)
        Dl.append(D)
        Il.append(np.where(I >= 0, I + lo, -1))
    Dm, Im = od.topk_merge(metric, np.stack(Dl), np.stack(Il))
    assert np.array_equal(Im, Iw) and np.array_equal(Dm, Dw)


# ---- sharding behind the drop-in API: VectorStorage.shard() / load(shard=...) -----------------------------------

def _store_worker(rank, world, port, tmpdir, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        warnings.simplefilter("ignore")
        from _data import zipf_queries
        from _oracle_backends import OracleFlatBackendF32, oracle_backends
        from test_host_logic import STRATEGIES, _build_store
        from rag_system_amd import faiss_compat
        from rag_system_amd.search_engine import SearchEngine
        from rag_system_amd.sharded import ShardContext
        from rag_system_amd.vector_storage import VectorStorageManager
        with oracle_backends():
            faiss_compat.set_backend_factory(OracleFlatBackendF32)     # float32 ties by id, like the kernels
            # every rank builds the same store (SPMD), keeps a whole copy as the reference and shards the other
            mgr, whole, em = _build_store(n_docs=50, seed=5, dim=48)
            _m2, store, _em = _build_store(n_docs=50, seed=5, dim=48)
            ctx = ShardContext()
            assert (ctx.rank, ctx.world) == (rank, world)
            store.shard(ctx)
            n = len(whole.chunks_data)
            assert store.index.ntotal == whole.index.ntotal == n and store.index.local.ntotal < n
            assert store.bm25.local.n_local < len(whole.bm25_corpus) and store.bm25.idf == whole.bm25.idf
            queries = [" ".join(t) for t in zipf_queries(5, 400, seed=9)] + ["zzzz qqqq", whole.chunks_data[5]["chunk"][:80]]
            # the three searches under the strategies: dense top-k, BM25 top-k and scores, TF-IDF top-k / ranks / scores
            for q in queries:
                qv = em.embed_query(q)
                for k in (1, 7, 40, n + 5):                       # incl. k > ntotal (padding: the last chunk quirk)
                    assert store.search(qv, k) == whole.search(qv, k), (rank, q, k)
                    assert store.search_bm25(q, k) == whole.search_bm25(q, k), (rank, q, k)
                from rag_system_amd.text import word_tokenize
                toks = word_tokenize(q.lower())
                assert store.bm25.get_scores(toks).tobytes() == whole.bm25.get_scores(toks).tobytes()
                ts, tw = store.get_tfidf_scorer(), whole.get_tfidf_scorer()
                rows = [0, 3, 7, len(whole.tfidf_doc_ids) - 1, 3]    # rows of both shards, one of them twice
                a, b = ts.query(q, 9, rows), tw.query(q, 9, rows)
                for x, y in zip(a, b):
                    assert np.asarray(x).tolist() == np.asarray(y).tolist(), (rank, q)
                assert ts.scores(q).tobytes() == tw.scores(q).tobytes()
            # every strategy through SearchEngine: the sharded store answers like the whole one
            es, ew = SearchEngine(store, em), SearchEngine(whole, em)
            for q in queries[:4]:
                for strategy, mode in STRATEGIES:
                    for kw in ({}, {"use_mmr": True, "mmr_reuse_index_vectors": True}):
                        got = es.search_with_strategy(q, strategy, 6, mode, **kw)
                        exp = ew.search_with_strategy(q, strategy, 6, mode, **kw)
                        assert got == exp, (rank, q, strategy, mode, kw)
            # MMR on the stored rows: the candidates' vectors are gathered from their owners, no re-encoding
            n_enc = []
            real_encode = em.model.encode
            em.model.encode = lambda texts, *a, **k: (n_enc.append(0 if isinstance(texts, str) else len(texts)),
                                                      real_encode(texts, *a, **k))[1]
            for q in queries[:3]:
                got = es.search_with_strategy(q, "faiss", 8, "chunk", use_mmr=True, mmr_lambda=0.7, mmr_reuse_index_vectors=True)
                assert got == ew.search_with_strategy(q, "faiss", 8, "chunk", use_mmr=True, mmr_lambda=0.7,
                                                      mmr_reuse_index_vectors=True)
            em.model.encode = real_encode
            assert all(c <= 1 for c in n_enc), n_enc      # only single query texts were encoded
            rows = [0, n - 1, n // 2, 3]
            assert store.index.gather_rows(rows).tobytes() == whole.index.reconstruct_n()[rows].tobytes()
            # rows added later go to the last rank; ids stay those of the unsharded store
            for st_ in (store, whole):
                st_.add_vector(em.embed_query("fresh words added later"), "late", "fresh words added later", "late.txt")
            qv = em.embed_query("fresh words added later")
            assert store.search(qv, 3) == whole.search(qv, 3) and store.search(qv, 1)[0]["doc_id"] == "late"
            with pytest.raises(ValueError):
                store.save(os.path.join(tmpdir, "nope"), em)
            # load(shard=...): only this rank's rows are read from index.faiss
            if rank == 0:
                mgr.save_storage("s", tmpdir, em, reencode=False)
            dist.barrier()
            loaded = VectorStorageManager().load_storage("s", tmpdir, bm25_params="config", shard=True)
            ref = VectorStorageManager().load_storage("s", tmpdir, bm25_params="config", shard=False)
            assert loaded.index.sharded and loaded.index.local.ntotal < ref.index.ntotal == loaded.index.ntotal, (
                loaded.index.local.ntotal, ref.index.ntotal, loaded.index.ntotal)
            auto = VectorStorageManager().load_storage("s", tmpdir)      # Config.ENGINE_SHARD = "auto": a group exists
            assert getattr(auto.index, "sharded", False)
            el, er = SearchEngine(loaded, em), SearchEngine(ref, em)
            for q in queries[:3]:
                for strategy, mode in STRATEGIES:
                    assert el.search_with_strategy(q, strategy, 5, mode) == er.search_with_strategy(q, strategy, 5, mode)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_vector_storage_answers_like_the_whole_store(world, tmp_path):
    """the multi-GPU path behind VectorStorage / SearchEngine, dense AND sparse: every rank of
    a gloo group holds a block of the dense rows and of the BM25 / TF-IDF postings (whole-corpus statistics) and
    returns, for every search and every strategy, exactly what the unsharded store returns -- ties across the
    shard border (the duplicated document of _build_store), k beyond the corpus, rows added after sharding."""
    port = 29500 + (os.getpid() % 2000) + 10 + world
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_store_worker, args=(world, port, str(tmp_path), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}
