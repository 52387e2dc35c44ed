#!/usr/bin/env python3
"""SDPA backends on MI355X for the DETR shapes of the SAM3 image model (bf16, fwd and fwd+bwd):
   fusion-encoder self-attention  q,k,v [8, 8, 5184, 32]
   decoder image cross-attention  q [8, 8, 401, 32], k,v [8, 8, 5184, 32] with an additive bias [8, 8, 401, 5184]
   and the same with the head dimension zero-padded to 64."""
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

dev = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n


import sys
P_DROP = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0      # attention-probability dropout (the model trains with 0.1)
print(f"dropout_p = {P_DROP}")
for name, (B, H, Lq, Lk, D, bias) in {"encoder self   [8,8,5184,32]": (8, 8, 5184, 5184, 32, False),
                                      "encoder self   pad d=64": (8, 8, 5184, 5184, 64, False),
                                      "decoder cross  [8,8,401x5184,32]+bias": (8, 8, 401, 5184, 32, True),
                                      "decoder cross  pad d=64 +bias": (8, 8, 401, 5184, 64, True)}.items():
    # nn.MultiheadAttention hands SDPA [B, H, L, D] views of [L, B*H, D] storage
    mk = lambda L: torch.randn(L, B * H, D, device=dev, dtype=torch.bfloat16).transpose(0, 1).reshape(B, H, L, D).requires_grad_(True)
    q, k, v = mk(Lq), mk(Lk), mk(Lk)
    m = torch.randn(B, H, Lq, Lk, device=dev, dtype=torch.bfloat16) if bias else None
    scale = 32 ** -0.5
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
        try:
            with sdpa_kernel(be):
                def run():
                    o = F.scaled_dot_product_attention(q, k, v, attn_mask=m, scale=scale, dropout_p=P_DROP)
                    o.backward(o)

                def fwd():
                    with torch.no_grad():
                        F.scaled_dot_product_attention(q, k, v, attn_mask=m, scale=scale, dropout_p=P_DROP)
                tf, tb = timeit(fwd), timeit(run)
                print(f"{name:40s} {be.name:20s} fwd {tf:7.3f} ms   fwd+bwd {tb:7.3f} ms")
        except Exception as e:
            print(f"{name:40s} {be.name:20s} failed: {str(e)[:90]}")
