#!/usr/bin/env python3
"""bench.py -- headline benchmark: QPS of exact IP top-10 over 10M x 384 fp32 (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W] [--nq 1] [--rows 10000000 --dim 384 --k 10]

One step = one pass of the hot path over one query batch: rag_flat_search_dev (pad + scan +
merge kernels) [+ RCCL all-gather of the per-rank top-k + rag_topk_merge_strided_dev when N > 1],
the (nq x k) result on the host, one stream synchronise -- every step, so a step is also the
latency of one query batch.  Everything is enqueued on ONE explicit (non-default) stream.  The
result reaches the host either because the last kernel writes it straight into pinned host
memory (`--out pinned`, the default: no copy) or by one D2H copy on the same stream
(`--out device`).  The corpus is resident in HBM before the timed region; the query batch
(nq*d*4 bytes) is resident too.

N > 1: launched by torch.distributed.run, one rank per GPU.  STRONG scaling: the same
10M-row corpus is row-sharded over the ranks (rank r holds rows [r*ceil(n/N), ...)), every
rank scans its shard for the same queries, candidates are exchanged with one all-gather
(the path's only exchange step) and merged on every rank.  The line then also
carries `stages` (per-rank scan / all-gather / merge milliseconds) and a `c4` leg: BASELINE
config 4, 100M x 768 fp32 split over the ranks (north_star's scaling workload).

N = 1 appends time-bounded legs for the other BASELINE configs beside the headline `value`
(`c2`, `c4_shard`, `batch1024`, `prefilter`, `c3`, `c5` -- the HBM-bound ones first, the matrix-heavy ones last:
DESIGN.md section 6), each with its own roofline; `--no-legs` skips them.

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

T0 = time.perf_counter()


def log(msg):
    if os.environ.get("RANK", "0") == "0":
        print("[bench %6.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


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
    ap.add_argument("--out", choices=["pinned", "device"], default="pinned",
                    help="where the last kernel writes the (nq x k) result: pinned host memory (no copy) "
                         "or HBM followed by one D2H copy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (A/B runs)")
    ap.add_argument("--dump-result", default=None, help="write the last step's (D, I) to this .npz (rank 0)")
    ap.add_argument("--no-batch-leg", action="store_true",
                    help="skip the extra batch-1024 measurement reported beside the batch-1 headline")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary-config legs (c2, c3, c4, c5)")
    ap.add_argument("--legs", default="c2,c3,c4_shard,c5", help="which secondary legs to run at N=1")
    ap.add_argument("--prefilter", action="store_true",
                    help="A/B: enable the opt-in bfloat16 prefilter (rag_flat_prefilter) on the index of the main run")
    ap.add_argument("--c4-rows", type=int, default=100_000_000, help="rows of the N>1 c4 leg (x 768 fp32)")
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


def cpu_baseline_batch(args, Q, nq=1024, sample_rows=400_000, budget_s=8.0):
    """The batch half of the metric on the host cores: numpy `Q @ X.T` (the BLAS sgemm faiss-cpu uses for
    nq >= 20, all cores) + a per-block argpartition top-k, on a bounded sample of the rows, scaled by the row
    ratio.  A restatement, not faiss."""
    from oracle import dense as od
    import numpy as np
    rows = min(sample_rows, args.rows)
    X = od.synth_rows(args.seed, 0, rows, args.dim)
    Q = np.ascontiguousarray(Q[:nq])
    k = args.k
    blk = 50_000

    def once():
        best_s = np.full((len(Q), k), -np.inf, dtype=np.float32)
        best_i = np.zeros((len(Q), k), dtype=np.int64)
        for b0 in range(0, rows, blk):
            S = Q @ X[b0:b0 + blk].T
            idx = np.argpartition(S, -k, axis=1)[:, -k:]
            cs = np.concatenate([best_s, np.take_along_axis(S, idx, axis=1)], axis=1)
            ci = np.concatenate([best_i, idx + b0], axis=1)
            keep = np.argpartition(cs, -k, axis=1)[:, -k:]
            best_s, best_i = np.take_along_axis(cs, keep, axis=1), np.take_along_axis(ci, keep, axis=1)
        return best_s, best_i
    once()
    reps, t0 = 0, time.perf_counter()
    while reps < 20:
        once()
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    return {"value": len(Q) * reps / el * rows / float(args.rows), "unit": "QPS", "cores": cores, "kind": "port",
            "sample": "nq=%d top-%d over the first %d of %d rows, numpy sgemm + argpartition in blocks of %d rows (%d "
                      "reps, %.1f s, %.0f GFLOP/s); QPS scaled by the row ratio; restatement, not faiss"
                      % (len(Q), k, rows, args.rows, blk, reps, el, 2.0 * len(Q) * rows * args.dim * reps / el / 1e9)}


def kernel_name(nq, k, n_local, bf, d=0):
    """Which kernel makes the timed pass (the engine's dispatch, csrc/flat.hip search_device; the
    rocprofv3 summaries under profiles/ show the same names)."""
    if k > 128:
        return "flat_stream_kernel (256-entry lists)" if nq <= 4 else "flat_scan_kernel (pairs)"
    if nq <= 8 or (nq <= 16 and n_local < (1 << 19)):
        return "flat_stream_kernel"
    if nq <= 192 and n_local >= (1 << 19) and (nq <= 64 or not bf):
        return "flat_wide_kernel (sample + filter passes%s)" % (
            "" if nq <= 64 else ", %d query chunks" % ((nq + 63) // 64))
    # rows of whole 128-byte K stages: bf16 from 65 queries (k >= 16 up to 128 queries), fp32 from 129; every k above 128
    if (d * (2 if bf else 4)) % 128 == 0 and ((bf and 64 < nq <= 128 and k >= 16) or nq > 128):
        return "flat_pipe_kernel (filter flow: sample pass + filter pass)"
    return "flat_mfma_kernel"


def roofline(nq, k, n_local, d, bf, n_launch, scan_ms, bytes_per_launch, steps):
    avg_ms = scan_ms / max(n_launch, 1)
    launches_per_step = n_launch / float(max(steps, 1))
    mfma_peak = BF16_MFMA_PEAK_TF if bf else FP32_MFMA_PEAK_TF
    esz = 2 if bf else 4
    # arithmetic intensity 2*nq/esz flop per byte against the ridge peak_flops / peak_bytes
    hbm_bound = 2.0 * nq / esz <= mfma_peak * 1e3 / HBM_PEAK_GBS
    if hbm_bound:
        ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None}
    else:
        flops = 2.0 * nq * n_local * d / (launches_per_step if launches_per_step else 1.0)
        ach = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        roof = {"bound": "mfma", "achieved": ach, "peak": mfma_peak, "unit": "TFLOP/s", "frac": ach / mfma_peak,
                "traffic": None}
    roof.update({"kernel": kernel_name(nq, k, n_local, bf, d), "avg_launch_ms": avg_ms, "launches": n_launch,
                 "algorithmic_bytes_per_launch": bytes_per_launch})
    return roof


class Runner:
    """One resident index + one query batch + the step function, on one explicit stream."""

    def __init__(self, env, rows_total, d, k, nq, storage, seed, out_mode, index=None):
        import torch
        from rag_system_amd import faiss_compat as faiss
        from rag_system_amd.sharded import ShardedFlatSearcher
        self.env, self.k, self.nq, self.d = env, k, nq, d
        self.rows_total = rows_total
        world, rank = env["world"], env["rank"]
        per = (rows_total + world - 1) // world
        self.base = rank * per
        self.n_local = max(0, min(rows_total, self.base + per) - self.base)
        self.bf = storage == "bf16"
        if index is None:
            index = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT, storage=storage)
            index._b.reserve(self.n_local)
            index._b.add_synthetic(self.n_local, seed, self.base)
        self.index = index      # resident corpus (shared between runners of different batch sizes)
        qgen = faiss.IndexFlatIP(d)
        qgen._b.add_synthetic(nq, 4321, 0)
        self.Q = qgen.reconstruct_n(0, nq)
        del qgen
        dev, stream = env["dev"], env["stream"]
        self.stream = stream
        with torch.cuda.stream(stream):
            self.q_dev = torch.from_numpy(self.Q).to(dev)
        words = ShardedFlatSearcher.words(nq, k)
        self.out_host = torch.zeros(words, dtype=torch.int64).pin_memory()   # ids + scores
        self.out_mode = out_mode

        ptrs = {}

        def local_search(q, kk, D_loc, I_loc, st):
            key = (id(q), id(D_loc))
            if key not in ptrs:      # the tensors are resident: their addresses are looked up once
                ptrs[key] = (q.data_ptr(), D_loc.data_ptr(), I_loc.data_ptr())
            qp, dp, ip = ptrs[key]
            self.index._b.search_dev(nq, qp, kk, dp, ip, self.base, st)

        self.searcher = ShardedFlatSearcher(local_search, 0, nq, k, dev, stream=stream,
                                            out=self.out_host if out_mode == "pinned" else None,
                                            force_exchange=env.get("force_exchange", False))
        self.D_host, self.I_host = self.searcher.unpack(self.out_host)

    def step(self):
        import torch
        self.searcher.search(self.q_dev)
        if self.out_mode != "pinned":
            with torch.cuda.stream(self.stream):
                self.out_host.copy_(self.searcher.out_packed, non_blocking=True)
        self.stream.synchronize()

    def fence(self):
        import torch
        import torch.distributed as dist
        if self.env["world"] > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps, warmup):
        """-> (elapsed seconds [max over ranks], n scan launches, their total ms, bytes per launch)."""
        import torch
        import torch.distributed as dist
        for _ in range(warmup):
            self.step()
        self.index._b.profiling(True)
        self.index._b.timing()  # drop anything recorded so far
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        elapsed = time.perf_counter() - t0
        self.index._b.profiling(False)
        n_launch, scan_ms, bpl = self.index._b.timing()
        if self.env["world"] > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.env["dev"])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, n_launch, scan_ms, bpl

    def pipelined(self, steps, depth=4):
        """Throughput with `depth` searches in flight (N = 1): a step's result goes to its own pinned
        slot and the host only waits for the search that used the slot `depth` steps earlier -- what a
        server that does not block on every single query sees.  Reported beside `value`, never as it."""
        import torch
        words = self.searcher.words(self.nq, self.k)
        m = self.nq * self.k
        slots = [torch.zeros(words, dtype=torch.int64).pin_memory() for _ in range(depth)]
        ptrs = [(s[m:].data_ptr(), s[:m].data_ptr()) for s in slots]     # (scores, ids) as in unpack()
        evs = [torch.cuda.Event() for _ in range(depth)]
        qp, st = self.q_dev.data_ptr(), self.stream.cuda_stream
        b = self.index._b

        def run(n):
            for i in range(n):
                s = i % depth
                if i >= depth:
                    evs[s].synchronize()
                b.search_dev(self.nq, qp, self.k, ptrs[s][0], ptrs[s][1], self.base, st)
                evs[s].record(self.stream)
        run(2 * depth)
        self.fence()
        t0 = time.perf_counter()
        run(steps)
        self.fence()
        el = time.perf_counter() - t0
        D0, I0 = self.searcher.unpack(slots[(steps - 1) % depth])
        assert torch.equal(I0, self.I_host) and torch.equal(D0, self.D_host), "pipelined result differs"
        return {"depth": depth, "value": self.nq * steps / el, "unit": "QPS", "steps": steps,
                "ms_per_step": el / steps * 1e3,
                "note": "%d searches in flight, each into its own pinned result slot; the host waits only for the "
                        "slot it is about to reuse" % depth}

    def check_plausible(self):
        Dn, In = self.D_host.numpy(), self.I_host.numpy()
        assert ((In >= 0) & (In < self.rows_total)).all(), "ids out of range"
        assert (Dn[:, :-1] >= Dn[:, 1:]).all(), "scores not sorted best-first"
        return Dn, In

    def stages(self, steps=20):
        """Per-rank stage times gathered on rank 0 (list indexed by rank)."""
        import torch.distributed as dist
        mine = self.searcher.time_stages(self.q_dev, steps)
        if self.env["world"] == 1:
            return [mine]
        allr = [None] * self.env["world"]
        dist.all_gather_object(allr, mine)
        return allr

    def leg(self, name, steps, warmup, workload):
        el, nl, ms, bpl = self.timed(steps, warmup)
        out = {"workload": workload, "nq": self.nq, "k": self.k, "value": self.nq * steps / el, "unit": "QPS",
               "steps": steps, "ms_per_step": el / steps * 1e3,
               "roofline": roofline(self.nq, self.k, self.n_local, self.d, self.bf, nl, ms, bpl, steps)}
        if self.env["rank"] == 0:
            self.check_plausible()
        return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_gpus, argv, child_cmd=None):
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (WORLD_SIZE unset, N > 1): this
    process becomes a plain launcher.  It has not touched the GPU (no torch import, no engine call, and
    it never re-execs itself): it starts `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a CHILD process, passes the children's stderr through, relays rank 0's one JSON
    line to its own stdout and returns the child's exit code.  `child_cmd` (or the environment variable
    RAG_BENCH_LAUNCH_CMD, a JSON list) replaces the launcher command; tests/test_bench_launch_cpu.py
    uses it to run this branch without GPUs."""
    import subprocess
    if child_cmd is None and os.environ.get("RAG_BENCH_LAUNCH_CMD"):
        child_cmd = json.loads(os.environ["RAG_BENCH_LAUNCH_CMD"])
    if child_cmd is None:
        child_cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                     "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
                     "--master-port", str(free_port()), os.path.abspath(__file__)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    log("launching %d ranks: %s" % (n_gpus, " ".join(child_cmd + list(argv))))
    proc = subprocess.Popen(child_cmd + list(argv), stdout=subprocess.PIPE, env=env, text=True)
    line, other = None, []
    for raw in proc.stdout:              # stderr is inherited: progress lines stream through
        s = raw.strip()
        if s.startswith("{") and '"metric"' in s:
            try:
                json.loads(s)
                line = s
                continue
            except ValueError:
                pass
        other.append(raw)
    rc = proc.wait()
    for raw in other:                    # anything else a rank printed goes to stderr, never to stdout
        sys.stderr.write(raw)
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py launcher: the ranks exited 0 without printing a result line\n")
        rc = 1
    return rc


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            sys.exit(self_launch(args.gpus, sys.argv[1:]))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    from rag_system_amd import _lib

    # RAG_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo as the transport -- a functional
    # rehearsal of the N > 1 path on a one-GPU box (never a performance number)
    rehearsal = os.environ.get("RAG_BENCH_REHEARSAL") == "1"
    if rehearsal:
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("RAG_BENCH_REHEARSAL_TIMEOUT", "90")), exit=True)
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # RAG_BENCH_FORCE_EXCHANGE=1 (N = 1 only): a one-rank RCCL group, and every step takes the all-gather +
    # merge path of N > 1 -- what the exchange step costs on the real transport before a second GPU exists
    force_exchange = world == 1 and os.environ.get("RAG_BENCH_FORCE_EXCHANGE") == "1"
    if force_exchange:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("MASTER_PORT", "29533"),
                                rank=0, world_size=1, device_id=dev)
    lib = _lib.engine(dev_index)
    for o in args.opt:
        name, val = o.split("=")
        _lib.check(lib.rag_set_option(name.encode(), int(val)))

    env = {"world": world, "rank": rank, "dev": dev, "stream": torch.cuda.Stream(dev), "force_exchange": force_exchange}
    n, d, k, nq = args.rows, args.dim, args.k, args.nq
    bf = args.storage == "bf16"

    log("building the %d x %d %s index (%d rank(s))" % (n, d, args.storage, world))
    main_run = Runner(env, n, d, k, nq, args.storage, args.seed, args.out)
    if args.prefilter:
        main_run.index.set_prefilter(True)
    log("timing %d steps" % args.steps)
    elapsed, n_launch, scan_ms, bytes_per_launch = main_run.timed(args.steps, args.warmup)
    stages = main_run.stages() if (world > 1 or force_exchange) else None
    piped = main_run.pipelined(args.steps) if (world == 1 and args.out == "pinned" and not force_exchange) else None

    out = None
    if rank == 0:
        Dn, In = main_run.check_plausible()     # outside the timed region: a plausible top-k
        if args.dump_result:
            np.savez(args.dump_result, D=Dn, I=In)
        ms_per_step = elapsed / args.steps * 1e3
        qps = nq * args.steps / elapsed
        roof = roofline(nq, k, main_run.n_local, d, bf, n_launch, scan_ms, bytes_per_launch, args.steps)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath))
                if (tj.get("rows") == n and tj.get("dim") == d and tj.get("nq") == nq
                        and tj.get("storage", "f32") == args.storage):
                    # not measured in this run: the PMC counters need rocprofv3 around the process
                    roof["traffic"] = tj["hbm_bytes_per_launch"]
                    roof["traffic_source"] = ("profiles/traffic.json (rocprofv3 --pmc pass of this command, "
                                              "recorded earlier): " + str(tj.get("source")))
            except Exception:
                pass
        out = {
            "metric": "QPS, exact IP top-%d over %dM x %d %s, batch %d"
                      % (k, n // 1_000_000, d, "bf16" if bf else "fp32", nq),
            "value": qps, "unit": "QPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16" if bf else "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if rehearsal else "")
                    + (" (one-rank RCCL group, every step through the all-gather + merge path of N > 1)" if force_exchange else ""),
            "config": {"workload": "%d x %d %s L2-normalised synthetic rows (seed %d), IndexFlatIP top-%d, nq=%d"
                                   % (n, d, "bf16" if bf else "fp32", args.seed, k, nq),
                       "rows": n, "dim": d, "k": k, "nq": nq, "storage": args.storage,
                       "result": ("written to pinned host memory by the last kernel" if args.out == "pinned"
                                  else "HBM + one D2H copy") + ", one stream synchronise per step",
                       "parallelism": "rows sharded over %d GPU(s), RCCL all-gather + merge" % world
                                      if world > 1 else "1 GPU"},
            "roofline": roof,
        }
        if stages is not None:
            out["stages"] = {"per_rank": stages,
                             "note": "ms per step on each rank's stream; allgather_ms includes waiting for the slowest rank"}
        if piped is not None:
            out["pipelined"] = piped

    # ---- N > 1: the batch half on the sharded corpus, then BASELINE config 4 (north_star's scaling workload), rows
    # split over the ranks
    batch_leg = None
    if world > 1 and nq == 1 and not args.no_batch_leg:
        log("batch-1024 leg")
        rb = Runner(env, n, d, k, 1024, args.storage, args.seed, args.out, index=main_run.index)
        batch_leg = rb.leg("batch1024", 5, 1, "the headline corpus, nq=1024")
        del rb
    if world > 1 and not args.no_legs:
        del main_run
        torch.cuda.empty_cache()
        rows_c4 = args.c4_rows
        per_gb = rows_c4 * 768 * 4 / world / 1e9
        if per_gb <= 250.0:
            log("c4 leg: %d x 768 fp32 over %d ranks (%.1f GB per rank)" % (rows_c4, world, per_gb))
            r4 = Runner(env, rows_c4, 768, 10, 1, "f32", args.seed, args.out)
            leg = r4.leg("c4", 20, 3, "%d x 768 fp32 rows split over %d ranks, IndexFlatIP top-10, nq=1" % (rows_c4, world))
            leg["stages"] = {"per_rank": r4.stages(10)}
            leg["scaling"] = "strong"
            if out is not None:
                out["c4"] = leg
            del r4
        elif out is not None:
            out["c4"] = {"skipped": "%.0f GB per rank does not fit" % per_gb}

    # ---- N = 1: the other BASELINE configs, time-bounded, beside the headline.  Order matters on this part: for
    # about a second after a bf16-MFMA leg (prefilter nq = 1024, c3) it gives a short HBM-bound kernel ~5 % less
    # (1M x 384, nq = 1: 0.245 against 0.233 ms per launch, benchmarks/ab/run_r03_leg_order.sh), so the HBM-bound legs
    # run first and the matrix-heavy ones last; every leg is its own measurement on its own resident index.
    legs = [x for x in args.legs.split(",") if x] if (world == 1 and not args.no_legs and nq == 1 and not bf) else []
    if "c2" in legs:
        log("c2 leg: 1M x 384 fp32")
        r = Runner(env, 1_000_000, 384, 10, 1, "f32", args.seed, args.out)
        out["c2"] = {"nq1": r.leg("c2", 300, 20, "1M x 384 fp32, IndexFlatIP top-10, nq=1")}
        del r
    if "c4_shard" in legs:
        log("c4_shard leg: one rank's 12.5M x 768 fp32 shard of config 4 at 8 GPUs")
        r = Runner(env, 12_500_000, 768, 10, 1, "f32", args.seed, args.out)
        out["c4_shard"] = r.leg("c4_shard", 30, 5, "12.5M x 768 fp32 = one of 8 shards of 100M x 768, top-10, nq=1")
        del r
    if "c2" in legs:
        r = Runner(env, 1_000_000, 384, 10, 64, "f32", args.seed, args.out)
        out["c2"]["nq64"] = r.leg("c2", 100, 10, "1M x 384 fp32, IndexFlatIP top-10, nq=64")
        del r
    # The metric has two halves (batch 1 and batch 1024).  `value` is the batch-1 half; the other
    # is measured right after it on the same resident corpus (not part of `value`).
    if world == 1 and nq == 1 and not args.no_batch_leg:
        log("batch-1024 leg")
        rb = Runner(env, n, d, k, 1024, args.storage, args.seed, args.out, index=main_run.index)
        batch_leg = rb.leg("batch1024", 5, 1, "the headline corpus, nq=1024")
        del rb

    # Opt-in bfloat16 prefilter (rag_flat_prefilter; never part of `value`): the same corpus and queries with a
    # bfloat16 copy of the rows as candidate source; results are those of the fp32 rows (compared below).
    pf_leg = None
    if world == 1 and nq == 1 and not bf and not args.no_batch_leg and not args.prefilter and k <= 16:
        log("prefilter leg")
        pf_leg = prefilter_leg(env, main_run, elapsed / args.steps * 1e3, args)
    if rank == 0 and batch_leg is not None:
        out["batch1024"] = batch_leg
    if rank == 0 and pf_leg is not None:
        out["prefilter"] = pf_leg
    if legs:
        del main_run
        torch.cuda.empty_cache()
        if "c3" in legs:
            log("c3 leg: 10M x 768 bf16, nq=1024, k=100")
            r = Runner(env, 10_000_000, 768, 100, 1024, "bf16", args.seed, args.out)
            out["c3"] = r.leg("c3", 8, 2, "10M x 768 bf16 rows, IndexFlatIP top-100, nq=1024")
            del r
        torch.cuda.empty_cache()
        if "c5" in legs:
            try:
                out["c5"] = c5_leg()
            except Exception as e:   # the headline line must still be printed
                out["c5"] = {"error": repr(e)}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not bf:
            log("cpu baseline")
            from rag_system_amd import faiss_compat as faiss
            qg = faiss.IndexFlatIP(d)
            qg._b.add_synthetic(max(nq, 1), 4321, 0)
            out["cpu_baseline"] = cpu_baseline(args, qg.reconstruct_n(0, max(nq, 1)))
            if out.get("batch1024") is not None:
                try:
                    qb = faiss.IndexFlatIP(d)
                    qb._b.add_synthetic(1024, 4321, 0)
                    out["batch1024"]["cpu_baseline"] = cpu_baseline_batch(args, qb.reconstruct_n(0, 1024))
                except Exception as e:
                    out["batch1024"]["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
    if world > 1 or force_exchange:
        dist.destroy_process_group()


def prefilter_leg(env, main_run, main_ms, args):
    """The headline corpus searched through the opt-in bfloat16 prefilter: nq = 1, 64, 1024, each beside the
    plain fp32 search of the same queries (ids compared outside the timed regions)."""
    import numpy as np
    idx = main_run.index
    n, d, k = main_run.rows_total, main_run.d, main_run.k
    legs = {}
    plain = {1: (main_run.D_host.numpy().copy(), main_run.I_host.numpy().copy(), main_ms)}
    for nq_, steps in ((1, 100), (64, 30), (1024, 5)):
        r = Runner(env, n, d, k, nq_, "f32", args.seed, args.out, index=idx)
        if nq_ not in plain:
            idx.set_prefilter(False)
            el0, _, _, _ = r.timed(steps, 2)
            plain[nq_] = (r.D_host.numpy().copy(), r.I_host.numpy().copy(), el0 / steps * 1e3)
        idx.set_prefilter(True)
        el, nl, ms, bpl = r.timed(steps, 3)
        redone = idx._b.prefilter_redone()
        D1, I1 = r.check_plausible()
        D0, I0, ms0 = plain[nq_]
        same = I0 == I1
        assert np.allclose(D0, D1, rtol=1e-5, atol=2e-6), "prefilter scores differ from the fp32 search"
        assert same.mean() >= 0.999, "prefilter ids differ from the fp32 search"
        avg = ms / max(nl, 1)
        legs["nq%d" % nq_] = {
            "value": nq_ * steps / el, "unit": "QPS", "steps": steps, "ms_per_step": el / steps * 1e3,
            "plain_fp32_ms_per_step": ms0, "queries_redone_last_step": redone,
            "ids_identical_to_plain_search": float(same.mean()),
            "copy_pass": {"kernel": idx._b.last_kernel(), "avg_launch_ms": avg, "launches": nl,
                          "bytes_per_launch": bpl, "GB/s": bpl / (avg * 1e-3) / 1e9 if avg > 0 else 0.0,
                          "TFLOP/s": 2.0 * nq_ * n * d / (avg * 1e-3) / 1e12 if avg > 0 else 0.0}}
        del r
    idx.set_prefilter(False)
    return {"workload": "the headline corpus with rag_flat_prefilter on: candidates (64 / 104 per query) from a "
                        "bfloat16 copy of the rows (+%.2f GB of HBM), exact scores from the fp32 rows, per-query "
                        "proof that no other row can reach the top %d (else a device-side fp32 redo)" % (n * d * 2 / 1e9, k),
            "note": "opt-in, off in every other number of this line; not a reduced-precision result: D / I are "
                    "those of the fp32 rows (tests/test_prefilter_gpu.py)", **legs}


def c5_leg(n_queries=40):
    """BASELINE config 5: strategy=rrf + MMR over 1M synthetic documents, whole query through the host classes
    (benchmarks/hybrid_workload.py), parity-checked after the timed loop, the same queries through
    rag_system_amd.SearchEngine, and the CPU restatement beside them.  Runs benchmarks/bench_hybrid.py as a CHILD
    process: the leg overlaps three streams (dense pass, BM25 scoring, TF-IDF scoring), and in this process --
    which has created and destroyed a dozen streams by now -- HIP maps them onto its four hardware queues
    differently than in a process that serves such queries (0.42 against 0.40 ms per query, same kernels)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "benchmarks", "bench_hybrid.py"), "--queries", str(n_queries),
           "--search-engine", "--cpu-sample-docs", "40000", "--cpu-queries", "4", "--check-queries", "3"]
    log("c5 leg: " + " ".join(cmd[1:]))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        raise RuntimeError("benchmarks/bench_hybrid.py exited %d" % r.returncode)
    j = json.loads(line[-1])
    return {"workload": j["config"]["workload"], "value": j["value"], "unit": "QPS", "ms_per_query": j["ms_per_query"],
            "stage_ms": j["stage_ms"], "sparse_roofline": j["sparse_roofline"],
            "parity_checked_queries": j["parity"]["queries_checked"],
            "through_search_engine": j.get("through_search_engine"), "cpu_baseline": j.get("cpu_baseline"),
            "process": "child (benchmarks/bench_hybrid.py)"}


if __name__ == "__main__":
    main()
