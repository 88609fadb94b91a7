#!/usr/bin/env python3
"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this box for bfloat16 shapes next to
config C3's contraction (rows x 768 against 1024 queries): a yardstick for csrc/flat_mfma_pipe.h, which also has
to select the top k (the GEMM only writes its product).  python benchmarks/gemm_yardstick.py"""
import time

import torch

dev = torch.device("cuda", 0)


def run(m, n, k, reps=10, dt=torch.bfloat16):
    a = torch.randn(m, k, device=dev, dtype=dt)
    b = torch.randn(n, k, device=dev, dtype=dt)
    c = torch.empty(m, n, device=dev, dtype=dt)
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
    print(str(dt).replace("torch.", "") + " M=%d N=%d K=%d: %.3f ms, %.0f TFLOP/s (output %.2f GB)" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9, m * n * c.element_size() / 1e9),
          flush=True)


run(8192, 8192, 8192)
run(16384, 16384, 4096)
run(1_000_000, 1024, 768)      # a tenth of C3's rows: the same contraction, product written as bfloat16
run(2_000_000, 1024, 768, 5)
run(1_000_000, 256, 768)

# the fp32 half of the headline metric (10M x 384 fp32 rows against 1024 queries): a tenth of the rows
torch.backends.cuda.matmul.allow_tf32 = False
run(8192, 8192, 8192, 5, torch.float32)
run(1_000_000, 1024, 384, 5, torch.float32)
run(1_000_000, 64, 384, 10, torch.float32)
