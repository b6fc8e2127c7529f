#!/usr/bin/env python3
"""Yardstick: what the vendor fp32 GEMM (torch.mm -> rocBLAS/hipBLASLt) sustains on this chip, for GEMM shapes equivalent to
the conv layers (M = cout, N = pixels, K = taps*cin).  Not part of the product path."""
import time

import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
for name, M, N, K in (("square 8192", 8192, 8192, 8192), ("3x3 384->192 @8x128^2", 192, 131072, 3456),
                      ("3x3 96->96 @8x128^2", 96, 131072, 864), ("1x1 192->96 @8x128^2", 96, 131072, 192),
                      ("5x5 512->384 @8x16^2", 384, 2048, 12800)):
    a = torch.randn(M, K, device=dev)
    b = torch.randn(K, N, device=dev)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{name:28s} M={M:5d} N={N:6d} K={K:5d}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:6.1f} TF/s")
