"""rag_flat_search_dev_noalloc (include/rag_mi355x.h): the entry point for callers that cannot
allocate.  Called through the C ABI and compared with rag_flat_search_dev -- same kernels, so
the outputs must be bit-identical -- on two calls in a row, the second with a smaller input
(the static scratch array keeps the first call's contents), for fp32 / bf16 rows, IP and L2,
padded and unpadded row lengths; limits are rejected with a clear error."""
import ctypes as C

import numpy as np
import pytest

from oracle import dense as od

pytestmark = pytest.mark.gpu


def _rand_unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    return x


def _both(lib, _lib, torch, idx, Q, k, id_base, stream):
    """(D, I) of the allocating and of the no-allocation entry point for the same call."""
    dev = torch.device("cuda:0")
    nq = Q.shape[0]
    outs = []
    with torch.cuda.stream(stream):
        qd = torch.from_numpy(Q).to(dev)
        for fn in (lib.rag_flat_search_dev, lib.rag_flat_search_dev_noalloc):
            Dd = torch.full((nq, k), 7.0, dtype=torch.float32, device=dev)
            Id = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
            _lib.check(fn(idx._b._h, nq, C.c_void_p(qd.data_ptr()), k, C.c_void_p(Dd.data_ptr()),
                          C.c_void_p(Id.data_ptr()), id_base, C.c_void_p(stream.cuda_stream)), lib)
            stream.synchronize()
            outs.append((Dd.cpu().numpy(), Id.cpu().numpy(), idx._b.last_kernel()))
    # both entry points dispatch these shapes to the same kernel and launch shape (the allocating one lets
    # the kernel's last block merge the block lists for 1-2 queries; the no-allocation one always runs the
    # merge kernel, so that nothing in its scratch has to start at zero)
    name0 = outs[0][2].replace(" fused-merge", "")
    assert name0.startswith("stream list") and outs[1][2] == name0 + " (noalloc)", (outs[0][2], outs[1][2])
    return [o[:2] for o in outs]


@pytest.mark.parametrize("storage,metric,d", [("f32", 0, 384), ("f32", 1, 100), ("bf16", 0, 768), ("bf16", 1, 37)])
def test_noalloc_matches_allocating_entry_point_two_calls_in_a_row(engine, storage, metric, d):
    import torch
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    rng = np.random.default_rng(1000 + d)
    stream = torch.cuda.Stream(torch.device("cuda:0"))
    n = 120_000
    X = _rand_unit(rng, n, d)
    X[70_000] = X[5]                                   # exact tie: lower id first in both
    idx = faiss.IndexFlat(d, metric, storage=storage)
    idx.add(X)
    idx.search(X[:1], 1)                               # flush staged rows, compute L2 row norms (allocating path)
    rnd = od.bf16_round if storage == "bf16" else (lambda a: a)
    Xr = rnd(X)
    # first call: the largest shape the entry point serves; second call: smaller nq, k and id_base
    for nq, k, id_base in [(8, 128, 5_000_000_000), (2, 10, 0), (1, 1, 17)]:
        Q = _rand_unit(rng, nq, d)
        Q[0] = X[5]
        (Da, Ia), (Dn, In) = _both(lib, _lib, torch, idx, Q, k, id_base, stream)
        assert np.array_equal(In, Ia), (storage, metric, nq, k)
        assert np.array_equal(Dn.view(np.uint32), Da.view(np.uint32)), (storage, metric, nq, k)
        # and both are right (oracle on the stored values)
        Qr = rnd(Q)
        _, Ir, D64 = od.flat_search(Xr, Qr, k, metric)
        X64, Q64 = Xr.astype(np.float64), Qr.astype(np.float64)
        od.check_topk(Dn, In - id_base, D64, Ir, 1e-5, 4e-6,
                      lambda qi, ids: (X64[ids] @ Q64[qi]) if metric == 0 else ((X64[ids] - Q64[qi]) ** 2).sum(1))
    # a smaller INDEX right after (fewer blocks write lists: stale lists of the big call must not leak in)
    small = faiss.IndexFlat(d, metric, storage=storage)
    small.add(X[:2_000])
    small.search(X[:1], 1)
    Q = _rand_unit(rng, 3, d)
    (Da, Ia), (Dn, In) = _both(lib, _lib, torch, small, Q, 20, 0, stream)
    assert np.array_equal(In, Ia) and np.array_equal(Dn.view(np.uint32), Da.view(np.uint32))
    assert In.max() < 2_000


def test_noalloc_limits_and_errors(engine):
    import torch
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    d = 64
    idx = faiss.IndexFlatIP(d)
    idx.add(_rand_unit(rng, 5000, d))
    q = torch.zeros((16, d), dtype=torch.float32, device=dev)
    D = torch.zeros((16, 256), dtype=torch.float32, device=dev)
    I = torch.zeros((16, 256), dtype=torch.int64, device=dev)

    def call(h, nq, k):
        return lib.rag_flat_search_dev_noalloc(h, nq, C.c_void_p(q.data_ptr()), k, C.c_void_p(D.data_ptr()),
                                               C.c_void_p(I.data_ptr()), 0, None)
    # rows still staged on the host: flushing would allocate
    assert call(idx._b._h, 1, 10) == _lib.RAG_ERR_UNSUPPORTED and b"staged" in lib.rag_last_error()
    idx.search(np.zeros((1, d), np.float32), 1)
    assert call(idx._b._h, 9, 10) == _lib.RAG_ERR_INVALID_ARG and b"maximum 8" in lib.rag_last_error()
    assert call(idx._b._h, 1, 129) == _lib.RAG_ERR_INVALID_ARG
    assert call(idx._b._h, 1, 0) == _lib.RAG_ERR_INVALID_ARG
    assert call(None, 1, 10) == _lib.RAG_ERR_INVALID_ARG
    assert call(idx._b._h, 0, 10) == 0
    assert call(idx._b._h, 8, 128) == 0
    torch.cuda.synchronize()
    tiny = faiss.IndexFlatIP(d)
    tiny.add(_rand_unit(rng, 5, d))
    tiny.search(np.zeros((1, d), np.float32), 1)
    assert call(tiny._b._h, 1, 3) == _lib.RAG_ERR_UNSUPPORTED      # fewer than 16 rows
    empty = faiss.IndexFlatIP(d)
    assert call(empty._b._h, 2, 3) == 0                            # empty index: pads, like the allocating call
    torch.cuda.synchronize()
    assert (I.view(-1)[:6].cpu().numpy() == -1).all()               # (nq x k) packed at the start of the buffer
    l2 = faiss.IndexFlatL2(d)
    l2._b.add_synthetic(1000, 1, 0)                                 # rows in HBM, norms not computed yet
    assert call(l2._b._h, 1, 3) == _lib.RAG_ERR_UNSUPPORTED and b"norms" in lib.rag_last_error()


def test_noalloc_on_another_stream_waits_for_the_row_norms(engine):
    """the L2 row norms the no-allocation search reads are written by the EARLIER allocating search,
    asynchronously on that search's stream.  Issued right behind it on another stream with no synchronisation
    in between, the no-allocation search must still see them (it orders itself behind the handle's previous
    search: event record + wait, nothing allocated)."""
    import torch
    from rag_system_amd import _lib, faiss_compat as faiss
    lib = _lib.engine()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    d, n, k = 96, 400_000, 10
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    Q = _rand_unit(rng, 4, d)
    for trial in range(3):
        idx = faiss.IndexFlatL2(d)
        idx._b.add_synthetic(n, 50 + trial, 0)                      # rows in HBM, norms not computed yet
        qd = torch.from_numpy(Q).to(dev)
        Da = torch.zeros((4, k), dtype=torch.float32, device=dev)
        Ia = torch.zeros((4, k), dtype=torch.int64, device=dev)
        Dn, In = torch.zeros_like(Da), torch.zeros_like(Ia)
        torch.cuda.synchronize()
        args = lambda D, I, s: (idx._b._h, 4, C.c_void_p(qd.data_ptr()), k, C.c_void_p(D.data_ptr()),  # noqa: E731
                                C.c_void_p(I.data_ptr()), 0, C.c_void_p(s.cuda_stream))
        _lib.check(lib.rag_flat_search_dev(*args(Da, Ia, sa)), lib)          # computes 400k norms on stream A
        _lib.check(lib.rag_flat_search_dev_noalloc(*args(Dn, In, sb)), lib)  # reads them on stream B
        torch.cuda.synchronize()
        assert torch.equal(Ia, In) and torch.equal(Da.view(torch.int32), Dn.view(torch.int32)), trial
