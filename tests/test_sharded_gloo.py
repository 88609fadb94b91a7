"""world_size-2 (gloo, CPU) test of the multi-GPU path's host logic: shard bounds, global ids,
all-gather + merge == searching the whole corpus.  The local search and the merge are the CPU
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

            def local_search(q, kk, D_loc, I_loc):
                D, I, _ = od.flat_search(Xl, q.numpy(), kk, metric)
                D_loc.copy_(torch.from_numpy(D))
                I_loc.copy_(torch.from_numpy(np.where(I >= 0, I + lo, -1)))

            def merge(m, Da, Ia, Do, Io):
                D, I = od.topk_merge(m, Da.numpy(), Ia.numpy())
                Do.copy_(torch.from_numpy(D))
                Io.copy_(torch.from_numpy(I))

            s = ShardedFlatSearcher(local_search, metric, nq, k, torch.device("cpu"), merge=merge)
            D, I = s.search(torch.from_numpy(Q))
            Dw, Iw, _ = od.flat_search(X, Q, k, metric)
            assert np.array_equal(I.numpy(), Iw), (rank, metric)
            assert np.array_equal(D.numpy(), Dw), (rank, metric)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    from rag_system_amd.sharded import shard_bounds
    assert shard_bounds(10, 1) == [(0, 10)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 1), (2, 0), (2, 0)]
    assert sum(c for _, c in shard_bounds(100_000_000, 8)) == 100_000_000


@pytest.mark.timeout(180)
def test_two_rank_sharded_search_equals_whole():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
