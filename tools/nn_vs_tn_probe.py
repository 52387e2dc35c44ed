#!/usr/bin/env python3
"""Backward dX GEMMs of the MLP: gy @ W (NN, what autograd runs) vs F.linear(gy, W^T copy) (TN)."""
import torch
import torch.nn.functional as F
dev = "cuda"
M = 41472
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, (out_f, in_f) in {"fc2 dX (gy[M,1024] @ W2[1024,4736])": (1024, 4736), "fc1 dX (gy[M,4736] @ W1[4736,1024])": (4736, 1024),
                            "proj dX": (1024, 1024), "qkv dX": (3072, 1024)}.items():
    gy = (torch.randn(M, out_f, device=dev) * 0.02).bfloat16()
    W = (torch.randn(out_f, in_f, device=dev) * 0.02).bfloat16()
    Wt = W.t().contiguous()
    for rep in range(2):
        print(f"{name:40s} NN {t(lambda: gy @ W):7.1f} us   TN(copy) {t(lambda: F.linear(gy, Wt)):7.1f} us", flush=True)
