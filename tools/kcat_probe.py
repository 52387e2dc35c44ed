#!/usr/bin/env python3
"""Cost of K-concatenation ([x | t] @ [W^T ; sB]) in hipBLASLt: frozen GEMMs of the MLP with K and K+16."""
import torch
import torch.nn.functional as F
dev = "cuda"
M = 41472
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (K, N) in ((1024, 4736), (4736, 1024)):
    for pad in (0, 16, 64):
        x = torch.randn(M, K + pad, device=dev, dtype=torch.bfloat16)
        W = torch.randn(N, K + pad, device=dev, dtype=torch.bfloat16)
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        us = t(lambda: F.linear(x, W, b))
        # backward-style NN GEMM: gy[M, N(+pad as K)] @ W[N(+pad), K]
        gy = torch.randn(M, N + pad, device=dev, dtype=torch.bfloat16)
        W2 = torch.randn(N + pad, K, device=dev, dtype=torch.bfloat16)
        us2 = t(lambda: gy @ W2)
        print(f"fwd linear K={K}+{pad:2d} N={N}: {us:7.1f} us   bwd gy@W  K={N}+{pad:2d} N={K}: {us2:7.1f} us", flush=True)
