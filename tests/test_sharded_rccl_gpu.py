"""The exchange step of the sharded search on the REAL transport (RCCL = torch.distributed backend "nccl"),
rehearsed with one rank: a one-GPU box cannot hold two RCCL ranks, but a one-rank group still runs
init_process_group, all_gather_into_tensor on the searcher's stream, the packed-buffer views and
rag_topk_merge_strided_dev behind it -- everything of rag_system_amd/sharded.py's RCCL branch except a second
peer (world_size 2 over gloo: tests/test_sharded_gloo.py).  No counterpart in the reference (single process).
"""
import os
import socket

import numpy as np
import pytest

from oracle import dense as od

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(240, method="thread")
def test_one_rank_rccl_exchange_equals_local_search(engine):
    import torch
    import torch.distributed as dist
    from rag_system_amd import faiss_compat as faiss
    from rag_system_amd.sharded import ShardedFlatSearcher
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        n, d, k, base = 50_000, 96, 10, 7_000_000
        rng = np.random.default_rng(8)
        X = rng.standard_normal((n, d)).astype(np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        idx = faiss.IndexFlatIP(d)
        idx.add(X)
        stream = torch.cuda.Stream(dev)
        for nq in (1, 5, 40):
            Q = rng.standard_normal((nq, d)).astype(np.float32)
            with torch.cuda.stream(stream):
                q_dev = torch.from_numpy(Q).to(dev)

            def local_search(q, kk, D_loc, I_loc, st, nq=nq):
                idx._b.search_dev(nq, q.data_ptr(), kk, D_loc.data_ptr(), I_loc.data_ptr(), base, st)

            s = ShardedFlatSearcher(local_search, 0, nq, k, dev, stream=stream, force_exchange=True)
            assert s.exchange and s.world == 1 and not s._stage
            for _ in range(3):          # different steps reuse the gather buffers
                D, I = s.search(q_dev)
            stream.synchronize()
            _, Ir, D64 = od.flat_search(X, Q, k, 0)
            X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
            od.check_topk(D.cpu().numpy(), I.cpu().numpy() - base, D64, Ir, 1e-5, 2e-6,
                          lambda qi, ids: X64[ids] @ Q64[qi])
            t = s.time_stages(q_dev, 3)
            assert t["scan_ms"] > 0 and t["allgather_ms"] >= 0 and t["merge_ms"] > 0
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300, method="thread")
def test_sharded_vector_storage_over_a_one_rank_rccl_group(engine, tmp_path):
    """VectorStorage.shard() / load(shard=True) on the HIP engine with RCCL as the transport (one rank, every
    exchange forced through the collectives): dense search through rag_flat_search_dev + all_gather_into_tensor
    + rag_topk_merge_strided_dev, BM25 / TF-IDF through their top-k + rag_csc_count_ahead + all-gather /
    all-reduce of device tensors -- results equal to the unsharded store's, every strategy."""
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _data import zipf_queries
    from test_host_logic import STRATEGIES, _build_store
    from rag_system_amd.search_engine import SearchEngine
    from rag_system_amd.sharded import ShardContext
    from rag_system_amd.vector_storage import VectorStorageManager
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        mgr, whole, em = _build_store(n_docs=120, seed=31, dim=96)
        _m, store, _e = _build_store(n_docs=120, seed=31, dim=96)
        ctx = ShardContext(force_exchange=True)
        assert ctx.device.type == "cuda" and ctx.world == 1
        store.shard(ctx)
        queries = [" ".join(t) for t in zipf_queries(5, 400, seed=3)] + [whole.chunks_data[7]["chunk"][:60]]
        es, ew = SearchEngine(store, em), SearchEngine(whole, em)
        for q in queries:
            qv = em.embed_query(q)
            for k in (1, 10, 40):
                assert store.search(qv, k) == whole.search(qv, k)
                assert store.search_bm25(q, k) == whole.search_bm25(q, k)
            rows = [0, 5, 9, 5]
            a, b = store.get_tfidf_scorer().query(q, 12, rows), whole.get_tfidf_scorer().query(q, 12, rows)
            for x, y in zip(a, b):
                assert np.asarray(x).tolist() == np.asarray(y).tolist()
            for strategy, mode in STRATEGIES:
                assert es.search_with_strategy(q, strategy, 6, mode) == ew.search_with_strategy(q, strategy, 6, mode)
        mgr.save_storage("s", str(tmp_path), em, reencode=False)
        loaded = VectorStorageManager().load_storage("s", str(tmp_path), bm25_params="config", shard=ctx)
        assert loaded.index.sharded
        el = SearchEngine(loaded, em)
        for q in queries[:3]:
            assert el.search_with_strategy(q, "rrf", 5) == ew.search_with_strategy(q, "rrf", 5)
    finally:
        dist.destroy_process_group()
