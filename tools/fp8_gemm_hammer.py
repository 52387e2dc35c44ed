#!/usr/bin/env python3
"""
Is torch._scaled_mm (hipBLASLt's fp8 GEMM) bit-reproducible launch after launch at the shapes the fp8 frozen-W mode calls it with?  Each
shape is run ITERS times on the same operands; every output is compared with the first on the device (one flag, one sync at the end) and
its largest magnitude is tracked.  A rare race inside the library (its stream-K kernels exchange partial tiles through a workspace)
would show as a mismatch -- the soak's rare non-finite step has an astronomically large finite value appear in a
backward whose dgrad GEMMs are these calls.
    python tools/fp8_gemm_hammer.py [--iters 20000]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    ap.add_argument("--only", default="", help="substring of the shape names to run")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device=dev).manual_seed(5)
    M = 41472
    shapes = [("fc1 forward  e4m3 [M,1024] x W[4736,1024]^T", torch.float8_e4m3fn, 1024, 4736, True),
              ("fc2 forward  e4m3 [M,4736] x W[1024,4736]^T", torch.float8_e4m3fn, 4736, 1024, True),
              ("proj dgrad   e5m2 [M,1024] x W[1024,1024]", torch.float8_e5m2, 1024, 1024, False),
              ("qkv dgrad    e5m2 [M,3072] x W[3072,1024]", torch.float8_e5m2, 3072, 1024, False),
              ("fc1 dgrad    e5m2 [M,4736] x W[4736,1024]", torch.float8_e5m2, 4736, 1024, False)]
    out = {}
    for name, adt, K, N, bias in shapes:
        if args.only and args.only not in name:
            continue
        a = (torch.randn(M, K, device=dev, generator=g) * 50.0).to(adt)
        w = (torch.randn(N, K, device=dev, generator=g) * 50.0).to(torch.float8_e4m3fn)          # [N, K]: used as w.t() (column-major B)
        sa, sb = torch.full((1,), 3e-3, device=dev), torch.full((1,), 2e-3, device=dev)
        b = torch.randn(N, device=dev, generator=g).bfloat16() if bias else None
        ref = torch._scaled_mm(a, w.t(), scale_a=sa, scale_b=sb, bias=b, out_dtype=torch.bfloat16)
        mism = torch.zeros((), dtype=torch.bool, device=dev)
        worst = torch.zeros((), device=dev)
        for _ in range(args.iters):
            y = torch._scaled_mm(a, w.t(), scale_a=sa, scale_b=sb, bias=b, out_dtype=torch.bfloat16)
            mism |= (y != ref).any()
            worst = torch.maximum(worst, y.float().abs().max())
        torch.cuda.synchronize()
        out[name] = {"launches": args.iters, "any_output_differs_from_the_first": bool(mism), "largest_magnitude": float(worst),
                     "reference_largest_magnitude": float(ref.float().abs().max())}
        print(name, out[name], flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
