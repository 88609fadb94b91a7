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


# ------------------------------------------------------------------------------------------------------------
# Sharding behind the drop-in API: one process per GPU, every rank holds the
# SAME host metadata (chunks_data, documents: strings) and ITS block of the numeric indices -- a contiguous block
# of dense rows, and of the BM25 / TF-IDF postings the block of document rows.  Every call below is SPMD: all ranks
# make it with the same arguments and all ranks get the same result.


class ShardContext:
    """Which rank of how many this process is, and the torch.distributed group the exchange goes through.
    Without an initialised process group: one rank (the wrappers then behave like the plain indices)."""

    def __init__(self, group=None, device=None, force_exchange=False):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self.group = torch, dist, group
        # force_exchange: a ONE-rank group still goes through the collectives (the rehearsal of the real transport
        # on a one-GPU box, tests/test_sharded_rccl_gpu.py)
        self.force_exchange = bool(force_exchange)
        if dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            backend = dist.get_backend(group)
        else:
            self.rank, self.world, backend = 0, 1, None
        if device is None:      # RCCL moves device tensors, gloo host tensors
            device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.device = torch.device(device)

    def all_gather(self, arr):
        """numpy array (same shape and dtype on every rank) -> array [world, ...] (the path's one exchange step)."""
        import numpy as np
        arr = np.ascontiguousarray(arr)
        if self.world == 1 and not (self.force_exchange and self._dist.is_initialized()):
            return arr[None]
        torch = self._torch
        t = torch.from_numpy(arr).to(self.device)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=self.device)
        self._dist.all_gather_into_tensor(out.view(-1), t.reshape(-1), group=self.group)
        return out.cpu().numpy()

    def all_reduce_sum(self, arr):
        import numpy as np
        arr = np.ascontiguousarray(arr)
        if self.world == 1 and not (self.force_exchange and self._dist.is_initialized()):
            return arr
        t = self._torch.from_numpy(arr.copy()).to(self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()


def merge_topk_host(scores, ids, k, larger_is_better=True):
    """[G, nq, kk] candidate lists (ids < 0 = padding) -> ([nq, k] scores, [nq, k] ids): best score first, lower id
    first on ties -- the order of faiss's result lists and of Python's stable descending sort (vector_storage.py:180);
    short results are padded with id -1 / the worst score."""
    import numpy as np
    G, nq, kk = scores.shape
    s = np.moveaxis(scores, 0, 1).reshape(nq, G * kk)
    i = np.moveaxis(ids, 0, 1).reshape(nq, G * kk)
    D = np.empty((nq, k), dtype=scores.dtype)
    I = np.empty((nq, k), dtype=np.int64)
    worst = -np.inf if larger_is_better else np.inf
    if scores.dtype == np.float32:
        worst = -np.finfo(np.float32).max if larger_is_better else np.finfo(np.float32).max
    for q in range(nq):
        pad = i[q] < 0
        key = -s[q] if larger_is_better else s[q]
        order = np.lexsort((i[q], key, pad))[:k]
        order = order[~pad[order]]
        D[q, :len(order)] = s[q][order]
        I[q, :len(order)] = i[q][order]
        D[q, len(order):] = worst
        I[q, len(order):] = -1
    return D, I


class ShardedIndexFlat:
    """faiss.IndexFlat surface (ntotal, add, search, d, metric_type) over ONE rank's block of the rows.
    Global row ids = first_row + local row; rows added later go to the LAST rank (ids stay contiguous)."""
    sharded = True

    def __init__(self, local, first_row, ntotal, ctx):
        self.local, self.first_row, self._ntotal, self.ctx = local, int(first_row), int(ntotal), ctx
        self.d, self.metric_type, self.is_trained = local.d, local.metric_type, True
        self._searchers = {}

    @property
    def ntotal(self):
        return self._ntotal

    def add(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("add expects a (n, %d) float32 array, got %s" % (self.d, x.shape))
        if self.ctx.rank == self.ctx.world - 1:
            if self.local.ntotal == 0:
                self.first_row = self._ntotal       # an empty last block starts where the index ends
            self.local.add(x)
        self._ntotal += x.shape[0]

    def _device_searcher(self, nq, k):
        key = (nq, k)
        s = self._searchers.get(key)
        if s is None:
            torch = self.ctx._torch
            b = self.local._b

            def local_search(q, kk, D_loc, I_loc, st):
                b.search_dev(nq, q.data_ptr(), kk, D_loc.data_ptr(), I_loc.data_ptr(), self.first_row, st)
            s = ShardedFlatSearcher(local_search, self.metric_type, nq, k, self.ctx.device, group=self.ctx.group,
                                    stream=torch.cuda.Stream(self.ctx.device), force_exchange=self.ctx.force_exchange)
            if len(self._searchers) > 16:
                self._searchers.clear()
            self._searchers[key] = s
        return s

    def search(self, x, k):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("search expects a (nq, %d) float32 array, got %s" % (self.d, x.shape))
        k = int(k)
        if k <= 0:
            raise ValueError("k must be positive")
        nq = x.shape[0]
        if nq == 0:
            return np.empty((0, k), dtype=np.float32), np.empty((0, k), dtype=np.int64)
        if self.ctx.device.type == "cuda" and hasattr(self.local._b, "search_dev"):
            # the kernels' path: local top-k with global ids, ONE RCCL all-gather, device merge (ShardedFlatSearcher)
            torch = self.ctx._torch
            s = self._device_searcher(nq, k)
            with torch.cuda.stream(s.stream):
                q = torch.from_numpy(x).to(self.ctx.device, non_blocking=True)
                D, I = s.search(q)
                Dh, Ih = D.cpu(), I.cpu()
            return Dh.numpy(), Ih.numpy()
        D, I = self.local.search(x, k)
        I = np.where(I >= 0, I + self.first_row, -1)
        Da, Ia = self.ctx.all_gather(D), self.ctx.all_gather(I)
        return merge_topk_host(Da, Ia, k, larger_is_better=self.metric_type == 0)

    def reset(self):
        self.local.reset()
        self._ntotal, self.first_row = 0, 0

    def gather_rows(self, rows):
        """(len(rows), d) float32: the stored vectors of the given GLOBAL rows, each from the rank that holds it
        (one all-gather of len(rows) x d floats) -- what MMR on stored rows needs for its handful of candidates."""
        import numpy as np
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        out = np.zeros((len(rows), self.d), dtype=np.float32)
        mine = (rows >= self.first_row) & (rows < self.first_row + self.local.ntotal)
        for j in np.nonzero(mine)[0]:
            out[j] = self.local.reconstruct_n(int(rows[j] - self.first_row), 1)[0]
        parts = self.ctx.all_gather(np.concatenate([out, mine[:, None].astype(np.float32)], axis=1))
        owner = parts[:, :, -1].argmax(axis=0)
        return np.ascontiguousarray(parts[owner, np.arange(len(rows)), :-1])

    def reconstruct_n(self, n0=0, ni=None):
        raise NotImplementedError("a row-sharded index is searched and appended to; write_index / save() need the "
                                  "whole matrix: save from an unsharded store")


class ShardedBM25:
    """rank_bm25.BM25Okapi surface (get_scores, top_k, k1, b, idf, avgdl, corpus_size) over one rank's block of
    the chunk rows: the LOCAL index holds this rank's postings with the statistics (idf, avgdl) of the WHOLE corpus
    (bm25.BM25Okapi(..., shard=(first, count)))."""

    def __init__(self, local, first_row, n_total, ctx):
        self.local, self.first_row, self.corpus_size, self.ctx = local, int(first_row), int(n_total), ctx
        self.k1, self.b, self.epsilon = local.k1, local.b, local.epsilon
        self.idf, self.avgdl, self.average_idf = local.idf, local.avgdl, local.average_idf

    def get_scores(self, query):
        import numpy as np
        per = (self.corpus_size + self.ctx.world - 1) // self.ctx.world
        mine = np.zeros(per, dtype=np.float64)
        s = self.local.get_scores(query) if self.local.n_local else np.zeros(0)
        mine[:len(s)] = s
        allr = self.ctx.all_gather(mine)
        return np.concatenate([allr[r][:cnt] for r, (_lo, cnt) in enumerate(shard_bounds(self.corpus_size, self.ctx.world))])

    def top_k(self, query, k):
        import numpy as np
        k = min(int(k), self.corpus_size)
        if k <= 0:
            return np.empty(0, np.int64), np.empty(0, np.float64)
        D = np.full((1, k), -np.inf, dtype=np.float64)
        I = np.full((1, k), -1, dtype=np.int64)
        if self.local.n_local:
            ids, sc = self.local.top_k(query, k)
            D[0, :len(ids)] = sc
            I[0, :len(ids)] = np.asarray(ids, dtype=np.int64) + self.first_row
        Dm, Im = merge_topk_host(self.ctx.all_gather(D), self.ctx.all_gather(I), k)
        return Im[0], Dm[0]


class ShardedTfidfScorer:
    """tfidf.TfidfScorer surface (scores, top_k, query) over one rank's block of the document rows."""

    def __init__(self, local, first_row, n_docs, ctx, vectorizer):
        self.local, self.first_row, self.n_docs, self.ctx, self.vectorizer = local, int(first_row), int(n_docs), ctx, vectorizer

    def _n_local(self):
        return self.local.n_docs if self.local is not None else 0

    def scores(self, query):
        import numpy as np
        per = (self.n_docs + self.ctx.world - 1) // self.ctx.world
        mine = np.zeros(per, dtype=np.float64)
        if self._n_local():
            s = self.local.scores(query)
            mine[:len(s)] = s
        allr = self.ctx.all_gather(mine)
        return np.concatenate([allr[r][:cnt] for r, (_lo, cnt) in enumerate(shard_bounds(self.n_docs, self.ctx.world))])

    def top_k(self, query, k):
        return self.query(query, k, [])[:2]

    def query(self, query, k, doc_indices):
        """(top_idx, top_scores, ranks, scores) as TfidfScorer.query: the first k documents of the stable
        descending ranking of ALL documents, and rank / score of the given (global) document rows.  A document's
        rank is the sum over the shards of the rows standing ahead of its (score, row) pair (rag_csc_count_ahead)."""
        import numpy as np
        k = min(int(k), self.n_docs)
        rows = np.asarray(doc_indices, dtype=np.int64).reshape(-1)
        D = np.full((1, max(k, 1)), -np.inf, dtype=np.float64)
        I = np.full((1, max(k, 1)), -1, dtype=np.int64)
        nl = self._n_local()
        if nl:
            ids, sc = self.local.top_k(query, min(max(k, 1), nl))      # (leaves the score vector on the device)
            D[0, :len(ids)] = sc
            I[0, :len(ids)] = np.asarray(ids, dtype=np.int64) + self.first_row
        Dm, Im = merge_topk_host(self.ctx.all_gather(D), self.ctx.all_gather(I), max(k, 1))
        top_idx, top_sc = Im[0][:k], Dm[0][:k]
        if len(rows) == 0:
            return top_idx, top_sc, [], []
        # scores of the queried rows: each from its owner (one exchange), then every shard counts what it holds
        # ahead of every pair (one more)
        mine = (rows >= self.first_row) & (rows < self.first_row + nl)
        sc = np.zeros(len(rows), dtype=np.float64)
        if mine.any():
            sc[mine] = self.local.gather(rows[mine] - self.first_row)
        owned = self.ctx.all_gather(np.stack([sc, mine.astype(np.float64)]))
        scores = (owned[:, 0, :] * owned[:, 1, :]).sum(axis=0)
        cnt = self.local.count_ahead(scores, rows - self.first_row) if nl else np.zeros(len(rows), dtype=np.int64)
        ranks = self.ctx.all_reduce_sum(np.asarray(cnt, dtype=np.int64))
        return top_idx, top_sc, ranks, scores
