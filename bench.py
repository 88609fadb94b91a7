#!/usr/bin/env python3
"""bench.py -- headline benchmark: QPS of exact IP top-10 over 10M x 384 fp32 (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W] [--nq 1] [--rows 10000000 --dim 384 --k 10]

One step = one pass of the hot path over one query batch: rag_flat_search_dev (pad + scan +
merge kernels) [+ RCCL all-gather of the per-rank top-k + rag_topk_merge_dev when N > 1] +
copy of the (nq x k) result to the host + stream synchronise.  The corpus is resident in
HBM before the timed region; the query batch (nq*d*4 bytes) is resident too.

N > 1: launched by torch.distributed.run, one rank per GPU.  STRONG scaling: the same
10M-row corpus is row-sharded over the ranks (rank r holds rows [r*ceil(n/N), ...)), every
rank scans its shard for the same queries, candidates are exchanged with one all-gather
(the path's only exchange step) and merged on every rank.

The JSON line carries `roofline` (scan kernel, measured live with HIP events on the stream
the kernel runs on) and `cpu_baseline` (oracle/flat_ref.c:flat_search_cpu_baseline, a
faiss-cpu-style fp32 scan, rank 0 at N=1 only, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = vector rate
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nq", type=int, default=1, help="query batch size (1 = HBM-bound headline)")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--storage", choices=["f32", "bf16"], default="f32",
                    help="row storage (bf16: BASELINE config 3; rows and queries rounded to bfloat16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (A/B runs)")
    ap.add_argument("--dump-result", default=None, help="write the last step's (D, I) to this .npz (rank 0)")
    ap.add_argument("--no-batch-leg", action="store_true",
                    help="skip the extra batch-1024 measurement reported beside the batch-1 headline")
    return ap.parse_args()


def cpu_baseline(args, Q):
    """faiss-cpu-style scan on the host cores, bounded sample, scaled to the full workload."""
    from oracle import dense as od
    import numpy as np
    rows = min(args.cpu_sample_rows, args.rows)
    X = od.synth_rows(args.seed, 0, rows, args.dim)
    q1 = np.ascontiguousarray(Q[:1])
    od.cpu_baseline_search(X, q1, args.k, 1)  # warm-up / page-in
    reps, t0 = 0, time.perf_counter()
    while True:
        od.cpu_baseline_search(X, q1, args.k, 1)
        reps += 1
        el = time.perf_counter() - t0
        if el > 12.0 or reps >= 200:
            break
    qps_sample = reps / el
    scale = rows / float(args.rows)
    out = {
        "value": qps_sample * scale, "unit": "QPS", "cores": 1, "kind": "port",
        "sample": "nq=1 top-%d over the first %d of %d rows (%d reps, %.1f s, %.2f GB/s of one core); "
                  "QPS scaled by rows ratio; faiss-cpu scans batch-1 on one thread"
                  % (args.k, rows, args.rows, reps, el, qps_sample * rows * args.dim * 4 / 1e9),
    }
    # what all host cores could do with the same scan split over rows (not what faiss does)
    nt = min(od.num_threads(), 16)
    if nt > 1:
        od.cpu_baseline_search(X, q1, args.k, nt)
        reps2, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 4.0 and reps2 < 200:
            od.cpu_baseline_search(X, q1, args.k, nt)
            reps2 += 1
        out["all_cores"] = {"value": reps2 / (time.perf_counter() - t1) * scale, "cores": nt}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    from rag_system_amd import _lib, faiss_compat as faiss
    from rag_system_amd.sharded import ShardedFlatSearcher

    # RAG_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo as the transport -- a functional
    # rehearsal of the N > 1 path on a one-GPU box (never a performance number)
    rehearsal = os.environ.get("RAG_BENCH_REHEARSAL") == "1"
    if rehearsal:
        import faulthandler
        faulthandler.dump_traceback_later(60, exit=True)   # a stuck rehearsal names its line
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = _lib.engine(dev_index)
    for o in args.opt:
        name, val = o.split("=")
        _lib.check(lib.rag_set_option(name.encode(), int(val)))

    n, d, k, nq = args.rows, args.dim, args.k, args.nq
    per = (n + world - 1) // world
    base = rank * per
    n_local = max(0, min(n, base + per) - base)

    index = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT, storage=args.storage)
    index._b.reserve(n_local)
    index._b.add_synthetic(n_local, args.seed, base)
    qgen = faiss.IndexFlatIP(d)
    qgen._b.add_synthetic(nq, 4321, 0)
    Q = qgen.reconstruct_n(0, nq)
    del qgen

    q_dev = torch.from_numpy(Q).to(dev)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream

    def local_search(q, kk, D_loc, I_loc):
        index._b.search_dev(nq, q.data_ptr(), kk, D_loc.data_ptr(), I_loc.data_ptr(), base, st)

    searcher = ShardedFlatSearcher(local_search, 0, nq, k, dev, stream=st)
    out_host = torch.empty_like(searcher.out_packed, device="cpu").pin_memory()  # ids + scores, one copy
    D_host, I_host = searcher.unpack(out_host)

    def step():
        searcher.search(q_dev)
        out_host.copy_(searcher.out_packed, non_blocking=True)
        stream.synchronize()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    index._b.profiling(True)
    index._b.timing()  # drop anything recorded so far
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    index._b.profiling(False)
    n_launch, scan_ms, bytes_per_launch = index._b.timing()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The metric has two halves (batch 1 and batch 1024).  `value` is the batch-1 half; the other
    # is measured right after it on the same resident corpus (not part of `value`).
    batch_leg = None
    if nq == 1 and not args.no_batch_leg:
        nqb, steps_b = 1024, 5
        qg = faiss.IndexFlatIP(d)
        qg._b.add_synthetic(nqb, 4321, 0)
        qb_dev = torch.from_numpy(qg.reconstruct_n(0, nqb)).to(dev)
        del qg

        def local_search_b(q, kk, D_loc, I_loc):
            index._b.search_dev(nqb, q.data_ptr(), kk, D_loc.data_ptr(), I_loc.data_ptr(), base, st)

        sb = ShardedFlatSearcher(local_search_b, 0, nqb, k, dev, stream=st)
        hb = torch.empty_like(sb.out_packed, device="cpu").pin_memory()

        def step_b():
            sb.search(qb_dev)
            hb.copy_(sb.out_packed, non_blocking=True)
            stream.synchronize()

        step_b()
        index._b.profiling(True)
        index._b.timing()
        fence()
        tb = time.perf_counter()
        for _ in range(steps_b):
            step_b()
        fence()
        el_b = time.perf_counter() - tb
        index._b.profiling(False)
        nl_b, ms_b, _ = index._b.timing()
        if world > 1:
            t = torch.tensor([el_b], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_b = float(t.item())
        k_ms = ms_b / max(nl_b, 1)
        peak_b = BF16_MFMA_PEAK_TF if args.storage == "bf16" else FP32_MFMA_PEAK_TF
        ach_b = 2.0 * nqb * n_local * d / (k_ms * 1e-3) / 1e12
        batch_leg = {"nq": nqb, "value": nqb * steps_b / el_b, "unit": "QPS", "steps": steps_b,
                     "ms_per_step": el_b / steps_b * 1e3,
                     "roofline": {"bound": "mfma", "achieved": ach_b, "peak": peak_b, "unit": "TFLOP/s",
                                  "frac": ach_b / peak_b, "kernel": "flat_mfma_kernel", "avg_launch_ms": k_ms}}

    if rank == 0:
        # outside the timed region: the last step's result must be a plausible top-k
        Dn, In = D_host.numpy(), I_host.numpy()
        assert ((In >= 0) & (In < n)).all(), "ids out of range"
        assert (Dn[:, :-1] >= Dn[:, 1:]).all(), "scores not sorted best-first"
        if args.dump_result:
            np.savez(args.dump_result, D=Dn, I=In)
        ms_per_step = elapsed / args.steps * 1e3
        qps = nq * args.steps / elapsed
        avg_scan_ms = scan_ms / max(n_launch, 1)
        launches_per_step = n_launch / float(args.steps)
        flops = 2.0 * nq * n_local * d * (1.0 / launches_per_step if launches_per_step else 1.0)
        bf = args.storage == "bf16"
        mfma_peak = BF16_MFMA_PEAK_TF if bf else FP32_MFMA_PEAK_TF
        esz = 2 if bf else 4
        # arithmetic intensity 2*nq/esz flop per byte against the ridge peak_flops / peak_bytes
        hbm_bound = 2.0 * nq / esz <= mfma_peak * 1e3 / HBM_PEAK_GBS
        if hbm_bound:
            ach = bytes_per_launch / (avg_scan_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None}
        else:
            ach = flops / (avg_scan_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": ach / mfma_peak, "traffic": None}
        # which kernel made the timed pass (the engine's dispatch, csrc/flat.hip search_device; the
        # rocprofv3 summaries under profiles/ show the same names)
        if k > 128:
            kname = "flat_scan_kernel" if nq < (2 if bf else 3) else "flat_scan_kernel (pairs)"
        elif nq <= 8 or (nq <= 16 and n_local < (1 << 19)):
            kname = "flat_stream_kernel"
        elif nq <= 192 and n_local >= (1 << 19):
            kname = "flat_wide_kernel (sample + filter passes%s)" % ("" if nq <= 64 else ", %d query chunks" % ((nq + 63) // 64))
        else:
            kname = "flat_mfma_kernel"
        roof.update({"kernel": kname, "avg_launch_ms": avg_scan_ms, "launches": n_launch,
                     "algorithmic_bytes_per_launch": bytes_per_launch})
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath))
                if (tj.get("rows") == n and tj.get("dim") == d and tj.get("nq") == nq
                        and tj.get("storage", "f32") == args.storage):
                    roof["traffic"] = tj["hbm_bytes_per_launch"]
                    roof["traffic_source"] = tj.get("source")
            except Exception:
                pass
        out = {
            "metric": "QPS, exact IP top-%d over %dM x %d %s, batch %d"
                      % (k, n // 1_000_000, d, "bf16" if bf else "fp32", nq),
            "value": qps, "unit": "QPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16" if bf else "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if rehearsal else ""),
            "config": {"workload": "%d x %d %s L2-normalised synthetic rows (seed %d), IndexFlatIP top-%d, nq=%d"
                                   % (n, d, "bf16" if bf else "fp32", args.seed, k, nq),
                       "rows": n, "dim": d, "k": k, "nq": nq, "storage": args.storage,
                       "parallelism": "rows sharded over %d GPU(s), RCCL all-gather + merge" % world
                                      if world > 1 else "1 GPU"},
            "roofline": roof,
        }
        if batch_leg is not None:
            out["batch1024"] = batch_leg
        if world == 1 and not args.no_cpu_baseline and not bf:
            out["cpu_baseline"] = cpu_baseline(args, Q)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
