#!/usr/bin/env python3
"""Which SDPA backend is faster on MI355X for the trunk's two attention shapes (fwd + bwd, bf16)?"""
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

import sys
dev = "cuda"
if len(sys.argv) > 1:      # python tools/sdpa_probe.py ck|aotriton
    torch.backends.cuda.preferred_rocm_fa_library(sys.argv[1])
    print("preferred_rocm_fa_library ->", torch.backends.cuda.preferred_rocm_fa_library())
shapes = {"window [72,16,576,64]": (72, 16, 576, 64), "global [8,16,5184,64]": (8, 16, 5184, 64)}
for name, (B, H, L, D) in shapes.items():
    # [B, L, H, D] storage viewed as [B, H, L, D], as the trunk hands it to SDPA
    q, k, v = (torch.randn(B, L, H, D, device=dev, dtype=torch.bfloat16).transpose(1, 2).requires_grad_(True) for _ in range(3))
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
        try:
            with sdpa_kernel(be):
                def run():
                    o = F.scaled_dot_product_attention(q, k, v)
                    o.backward(o)
                def fwd():
                    with torch.no_grad():
                        F.scaled_dot_product_attention(q, k, v)
                res = []
                for fn in (fwd, run):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(10):
                        fn()
                    b.record(); b.synchronize()
                    res.append(a.elapsed_time(b) / 10)
                fl = 4 * B * H * L * L * D / 1e12
                print(f"{name:26s} {be.name:20s} fwd {res[0]:7.3f} ms {fl / res[0] * 1e3:6.1f} TF/s   fwd+bwd {res[1]:7.3f} ms "
                      f"{3.5 * fl / res[1] * 1e3:6.1f} TF/s")
        except Exception as e:
            print(f"{name:26s} {be.name:20s} failed: {str(e)[:80]}")
