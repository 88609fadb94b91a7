"""Row-sharded flat index over the GPUs of one node: one process per GPU.

SURVEY 8(e): rank r holds the contiguous row block [r*ceil(N/G), min(N, (r+1)*ceil(N/G)));
global id = block base + local row.  A search broadcasts nothing (every rank is given the
same query batch), runs the fused scan+top-k on its shard with id_base = block base, then the
per-rank (nq x k) candidate lists -- ids and scores packed in one buffer -- are exchanged with
ONE all-gather (RCCL over xGMI when the tensors are on the GPU: `backend="nccl"` is RCCL on
ROCm) and merged on every rank (rag_topk_merge_strided_dev: score best-first, lower global id
first on ties).  This is the path's only exchange step; there is no all-reduce.

Stream order: local search, all-gather and merge are all enqueued on ONE stream -- the
searcher's `stream` (a torch.cuda.Stream; None = whatever torch's current stream is at the
call) -- which is made torch's current stream for the duration of search(), so that
torch.distributed orders the collective behind the search and the merge behind the
collective.  local_search receives that stream's handle and must enqueue on it.

The reference has no counterpart (single process, faiss-cpu); the merged result is by
construction identical to searching one index that holds all rows (tests/test_flat_gpu.py
::test_sharded_search_and_device_merge, tests/test_sharded_gloo.py).
"""
import contextlib
import ctypes as C
import os

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
    [nq, k] through rag_topk_merge_strided_dev, asynchronous on `stream` (hipStream_t as int;
    0 = the null stream)."""
    lib = _lib.engine()
    G, nq, k = D_all.shape
    assert D_all.stride(1) == k and D_all.stride(2) == 1 and I_all.stride(1) == k and I_all.stride(2) == 1
    _lib.check(lib.rag_topk_merge_strided_dev(int(metric), int(G), int(nq), int(k), C.c_void_p(D_all.data_ptr()),
                                              C.c_void_p(I_all.data_ptr()), int(D_all.stride(0)),
                                              int(I_all.stride(0)), C.c_void_p(D_out.data_ptr()),
                                              C.c_void_p(I_out.data_ptr()), C.c_void_p(int(stream or 0))), lib)


class ShardedFlatSearcher:
    """Searches a row-sharded corpus with torch.distributed as the transport.

    local_search(q_tensor, k, D_loc, I_loc, stream_handle): fills the local candidate tensors with
        GLOBAL ids, asynchronously on stream_handle (int; 0 for CPU tensors / the null stream).
    merge(metric, D_all, I_all, D_out, I_out, stream_handle): merges the gathered lists.
    The defaults drive librag_mi355x.so; tests inject CPU implementations (gloo).
    out: optional int64 tensor of `words(nq, k)` elements that receives the packed result (ids,
        then scores) -- e.g. pinned host memory, which the GPU writes directly (no copy back).
    """

    @staticmethod
    def words(nq, k):
        m = nq * k
        return m + (m + 1) // 2

    def __init__(self, local_search, metric, nq, k, device, group=None, merge=None, stream=None, out=None,
                 force_exchange=False):
        import torch
        import torch.distributed as dist
        self._torch = torch
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.metric, self.nq, self.k = int(metric), int(nq), int(k)
        self.local_search = local_search
        self.merge = merge or _hip_merge
        self.stream = stream
        self._stream_handle = None
        self.on_gpu = torch.device(device).type == "cuda"
        # one packed buffer per rank: nq*k int64 ids, then nq*k float32 scores (padded to 8 bytes)
        m = nq * k
        words = self.words(nq, k)
        if out is not None:
            assert out.dtype == torch.int64 and out.numel() == words and out.is_contiguous()
        self._stage = False
        # force_exchange: take the all-gather + merge path with ONE rank too (a rehearsal of the collective's
        # stream ordering and buffer layout on a one-GPU box: tests/test_sharded_rccl_gpu.py)
        self.exchange = self.world > 1 or bool(force_exchange)
        if self.exchange:
            self._pack = torch.zeros(words, dtype=torch.int64, device=device)
            self._pack_all = torch.zeros((self.world, words), dtype=torch.int64, device=device)
            self.I_all = self._pack_all[:, :m].view(self.world, nq, k)
            self.D_all = self._pack_all[:, m:].view(torch.float32)[:, :m].view(self.world, nq, k)
            self.out_packed = out if out is not None else torch.zeros(words, dtype=torch.int64, device=device)
            # gloo cannot move GPU tensors by itself: torch's ProcessGroupGloo stages them through
            # pinned temporaries on side streams of its own and synchronises those from a worker
            # thread.  With several ranks on ONE card (the one-GPU rehearsal of bench.py) that path
            # takes 0.02-0.5 s per collective even with every stage ordered on one stream (measured
            # twice), so the packed lists are staged here instead, on the
            # searcher's stream: 0.6 ms per step.  RAG_GLOO_STAGE=0 selects torch's own path.
            # RCCL (the real transport) takes device tensors directly and never comes here.
            self._stage = (self.on_gpu and dist.is_initialized() and dist.get_backend(group) == "gloo"
                           and os.environ.get("RAG_GLOO_STAGE", "1") == "1")
            if self._stage:
                self._h_pack = torch.zeros(words, dtype=torch.int64).pin_memory()
                self._h_all = torch.zeros((self.world, words), dtype=torch.int64).pin_memory()
        else:
            self._pack = out if out is not None else torch.zeros(words, dtype=torch.int64, device=device)
            self.out_packed = self._pack
        self.D_loc, self.I_loc = self.unpack(self._pack)
        self.D_out, self.I_out = self.unpack(self.out_packed)
        self.stage_events = None   # set by time_stages()

    def unpack(self, packed):
        """(D, I) views of a tensor with out_packed's layout (e.g. its pinned host copy: ONE copy
        brings ids and scores back)."""
        m = self.nq * self.k
        return (packed[m:].view(self._torch.float32)[:m].view(self.nq, self.k),
                packed[:m].view(self.nq, self.k))

    def _handle(self):
        if not self.on_gpu:
            return 0
        if self.stream is not None:
            if self._stream_handle is None:
                self._stream_handle = self.stream.cuda_stream
            return self._stream_handle
        return self._torch.cuda.current_stream().cuda_stream

    def search(self, q):
        """q: (nq, d) tensor on the searcher's device.  Returns (D_out, I_out) tensors, complete when
        the searcher's stream reaches this point (GPU) / on return (CPU)."""
        torch = self._torch
        # one rank and no stage timing: nothing of torch's is enqueued here (local_search gets the stream's
        # handle explicitly), so the current-stream switch, a few microseconds per call, is skipped
        need_ctx = self.on_gpu and self.stream is not None and (self.exchange or self.stage_events)
        ctx = torch.cuda.stream(self.stream) if need_ctx else contextlib.nullcontext()
        with ctx:
            st = self._handle()
            ev = self.stage_events
            if ev:
                ev[0].record()
            self.local_search(q, self.k, self.D_loc, self.I_loc, st)
            if ev:
                ev[1].record()
            if self.exchange:
                if self._stage:
                    self._h_pack.copy_(self._pack, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                    self._dist.all_gather_into_tensor(self._h_all.view(-1), self._h_pack, group=self.group)
                    self._pack_all.copy_(self._h_all, non_blocking=True)
                else:
                    # torch orders the collective behind the work already on the current stream and
                    # the current stream behind the collective (async_op=False)
                    self._dist.all_gather_into_tensor(self._pack_all.view(-1), self._pack, group=self.group)
                if ev:
                    ev[2].record()
                self.merge(self.metric, self.D_all, self.I_all, self.D_out, self.I_out, st)
                if ev:
                    ev[3].record()
        return self.D_out, self.I_out

    def time_stages(self, q, steps):
        """Average milliseconds of the three stages of a search on THIS rank's stream (GPU only):
        local scan+select, all-gather (includes waiting for the slowest rank), merge.  Run apart
        from any timed loop."""
        torch = self._torch
        if not self.on_gpu:
            return None
        self.stage_events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = [0.0, 0.0, 0.0]
        try:
            for _ in range(steps):
                self.search(q)
                (self.stream or torch.cuda.current_stream()).synchronize()
                ev = self.stage_events
                acc[0] += ev[0].elapsed_time(ev[1])
                if self.exchange:
                    acc[1] += ev[1].elapsed_time(ev[2])
                    acc[2] += ev[2].elapsed_time(ev[3])
        finally:
            self.stage_events = None
        return {"scan_ms": acc[0] / steps, "allgather_ms": acc[1] / steps, "merge_ms": acc[2] / steps}
