"""Row-sharded flat index over the GPUs of one node: one process per GPU.

SURVEY 8(e): rank r holds the contiguous row block [r*ceil(N/G), min(N, (r+1)*ceil(N/G)));
global id = block base + local row.  A search broadcasts nothing (every rank is given the
same query batch), runs the fused scan+top-k on its shard with id_base = block base, then the
per-rank (nq x k) candidate lists -- ids and scores packed in one buffer -- are exchanged with
ONE all-gather (RCCL over xGMI when the tensors are on the GPU: `backend="nccl"` is RCCL on
ROCm) and merged on every rank (rag_topk_merge_strided_dev: score best-first, lower global id
first on ties).  This is the path's only exchange step; there is no all-reduce.

The reference has no counterpart (single process, faiss-cpu); the merged result is by
construction identical to searching one index that holds all rows (tests/test_flat_gpu.py
::test_sharded_search_and_device_merge, tests/test_sharded_gloo.py).
"""
import ctypes as C

import numpy as np

from . import _lib


def shard_bounds(n_total, world):
    """[(first_row, n_rows)] per rank."""
    per = (n_total + world - 1) // world if world > 0 else 0
    out = []
    for r in range(world):
        lo = min(n_total, r * per)
        hi = min(n_total, lo + per)
        out.append((lo, hi - lo))
    return out


def _hip_merge(metric, D_all, I_all, D_out, I_out, stream):
    """Device tensors [G, nq, k] (views into the packed gather buffer: list stride != nq*k) ->
    [nq, k] through rag_topk_merge_strided_dev."""
    lib = _lib.engine()
    G, nq, k = D_all.shape
    assert D_all.stride(1) == k and D_all.stride(2) == 1 and I_all.stride(1) == k and I_all.stride(2) == 1
    _lib.check(lib.rag_topk_merge_strided_dev(int(metric), int(G), int(nq), int(k), C.c_void_p(D_all.data_ptr()),
                                              C.c_void_p(I_all.data_ptr()), int(D_all.stride(0)),
                                              int(I_all.stride(0)), C.c_void_p(D_out.data_ptr()),
                                              C.c_void_p(I_out.data_ptr()),
                                              C.c_void_p(stream) if stream else None), lib)


class ShardedFlatSearcher:
    """Searches a row-sharded corpus with torch.distributed as the transport.

    local_search(q_tensor, k, D_loc, I_loc): fills the local candidate tensors with GLOBAL ids.
    merge(metric, D_all, I_all, D_out, I_out): merges the gathered lists.
    The defaults drive librag_mi355x.so; tests inject CPU implementations (gloo).
    """

    def __init__(self, local_search, metric, nq, k, device, group=None, merge=None, stream=None):
        import torch
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.metric, self.nq, self.k = int(metric), int(nq), int(k)
        self.local_search = local_search
        self.merge = merge or (lambda m, Da, Ia, Do, Io: _hip_merge(m, Da, Ia, Do, Io, stream))
        # one packed buffer per rank: nq*k int64 ids, then nq*k float32 scores (padded to 8 bytes)
        m = nq * k
        words = m + (m + 1) // 2
        self._pack = torch.zeros(words, dtype=torch.int64, device=device)
        self.I_loc = self._pack[:m].view(nq, k)
        self.D_loc = self._pack[m:].view(torch.float32)[:m].view(nq, k)
        if self.world > 1:
            self._pack_all = torch.zeros((self.world, words), dtype=torch.int64, device=device)
            self.I_all = self._pack_all[:, :m].view(self.world, nq, k)
            self.D_all = self._pack_all[:, m:].view(torch.float32)[:, :m].view(self.world, nq, k)
            self.out_packed = torch.zeros(words, dtype=torch.int64, device=device)
            self.I_out = self.out_packed[:m].view(nq, k)
            self.D_out = self.out_packed[m:].view(torch.float32)[:m].view(nq, k)
            # gloo with GPU tensors (the one-GPU functional rehearsal of bench.py; RCCL is the real
            # transport): gloo's own device path stalled every other launch with several ranks on one
            # card, so the packed lists are staged through host memory there
            self._stage = (torch.device(device).type == "cuda" and dist.is_initialized()
                           and dist.get_backend(group) == "gloo")
            if self._stage:
                self._h_pack = torch.zeros(words, dtype=torch.int64)
                self._h_all = torch.zeros((self.world, words), dtype=torch.int64)
        else:
            self.out_packed, self.D_out, self.I_out = self._pack, self.D_loc, self.I_loc

    def unpack(self, packed):
        """(D, I) views of a tensor with out_packed's layout (e.g. its pinned host copy: ONE copy
        brings ids and scores back)."""
        import torch
        m = self.nq * self.k
        return packed[m:].view(torch.float32)[:m].view(self.nq, self.k), packed[:m].view(self.nq, self.k)

    def search(self, q):
        """q: (nq, d) tensor on the searcher's device.  Returns (D_out, I_out) tensors (asynchronous
        on the current stream for GPU tensors)."""
        self.local_search(q, self.k, self.D_loc, self.I_loc)
        if self.world > 1:
            if self._stage:
                self._h_pack.copy_(self._pack)        # ordered behind the local search on the current stream
                self._dist.all_gather_into_tensor(self._h_all.view(-1), self._h_pack, group=self.group)
                self._pack_all.copy_(self._h_all)
            else:
                self._dist.all_gather_into_tensor(self._pack_all.view(-1), self._pack, group=self.group)
            self.merge(self.metric, self.D_all, self.I_all, self.D_out, self.I_out)
        return self.D_out, self.I_out
