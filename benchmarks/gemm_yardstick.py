#!/usr/bin/env python3
"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this box for bfloat16 shapes next to
config C3's contraction (rows x 768 against 1024 queries): a yardstick for csrc/flat_mfma_pipe.h, which also has
to select the top k (the GEMM only writes its product).  python benchmarks/gemm_yardstick.py"""
import time

import torch

dev = torch.device("cuda", 0)


def run(m, n, k, reps=10):
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.matmul(a, b.t(), out=c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("M=%d N=%d K=%d: %.3f ms, %.0f TFLOP/s (output %.2f GB)" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9, m * n * 2 / 1e9),
          flush=True)


run(8192, 8192, 8192)
run(16384, 16384, 4096)
run(1_000_000, 1024, 768)      # a tenth of C3's rows: the same contraction, product written as bfloat16
run(2_000_000, 1024, 768, 5)
run(1_000_000, 256, 768)
