#!/usr/bin/env python3
"""Attention forward at the trunk's two shapes (batch 8): this library's kernel vs PyTorch-ROCm's efficient / flash forward,
and forward + backward through both pairings.  Usage (GPU box): python tools/attn_bench.py > gpurun_out/attn_bench.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.nn.attention import SDPBackend, sdpa_kernel
import torch.nn.functional as F

from sam3_lora_amd.vit import _HipAttention, _attention

dev = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


out = {}
for name, (B, L, H) in {"windows [72, 576, 16, 64]": (72, 576, 16), "global [8, 5184, 16, 64]": (8, 5184, 16)}.items():
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(B, L, H, 64, device=dev, generator=g).bfloat16().requires_grad_(True) for _ in range(3))
    go = torch.randn(B, L, H, 64, device=dev, generator=g).bfloat16()
    flops = 4.0 * B * H * L * L * 64
    r = {"fwd_flop": flops}
    assert _HipAttention.usable(q, k, v)
    with torch.no_grad():
        t = timeit(lambda: _HipAttention._launch(q, k, v))
        r["hip_fwd_us"] = round(t, 1)
        r["hip_fwd_tflops"] = round(flops / t / 1e6, 1)
        for be, tag in ((SDPBackend.EFFICIENT_ATTENTION, "efficient"), (SDPBackend.FLASH_ATTENTION, "flash")):
            try:
                with sdpa_kernel([be]):
                    t = timeit(lambda: F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)))
                r[f"torch_{tag}_fwd_us"] = round(t, 1)
                r[f"torch_{tag}_fwd_tflops"] = round(flops / t / 1e6, 1)
            except Exception as e:
                r[f"torch_{tag}_fwd_us"] = f"{type(e).__name__}: {str(e)[:80]}"

    def fb(fn):
        def run():
            for x in (q, k, v):
                x.grad = None
            fn().backward(go)
        return run
    r["hip_fwd+torch_bwd_us"] = round(timeit(fb(lambda: _attention(q, k, v)), iters=10), 1)
    os.environ["SAM3_HIP_ATTENTION"] = "0"
    r["torch_fwd+bwd_us"] = round(timeit(fb(lambda: _attention(q, k, v)), iters=10), 1)
    os.environ.pop("SAM3_HIP_ATTENTION")
    out[name] = r
print(json.dumps(out, indent=1))
