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
