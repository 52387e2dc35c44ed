#!/usr/bin/env python3
"""The command profiled for k_fused_linear's PMC passes: 12 calls of sam3_lora_linear_fwd (+ GELU) at the fc1 site of
BASELINE configs[1] (M = 41,472, 1024 -> 4736, r = 16), nothing else on the device."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sam3_lora_amd import functional as Fn

if __name__ == "__main__":
    DEV = "cuda:0"
    M, fin, fout, rank, s = 41472, int(os.environ.get("PROBE_K", 1024)), 4736, 16, 2.0
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    W = (torch.randn(fout, fin, device=DEV, generator=g) / 32).bfloat16()
    b = (torch.randn(fout, device=DEV, generator=g) * 0.1).bfloat16()
    A = (torch.rand(fin, rank, device=DEV, generator=g) - 0.5) / 2
    B = torch.randn(rank, fout, device=DEV, generator=g) * 0.05
    blob = Fn.pack_operands(A, B, 0)
    h = torch.empty(M, fout, device=DEV, dtype=torch.bfloat16)
    a = torch.empty_like(h)
    for _ in range(12):
        Fn.lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=True, y_out=h, gelu_out=a)
    torch.cuda.synchronize()
