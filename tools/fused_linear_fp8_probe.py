#!/usr/bin/env python3
"""
The fc1 -> GELU site in the fp8 frozen-W mode at BASELINE configs[1]'s size (M = 8 x 5184, 1024 -> 4736, r = 16): hipBLASLt's fp8
GEMM (torch._scaled_mm) + the adapter pass with GELU and the fp8 image (sam3_lora_fwd_act_q8) against the ONE-kernel form
(sam3_lora_linear_fwd_q8), the bf16 fused kernel beside them; interleaved rounds in one process, in-situ kernel split.
usage: python tools/fused_linear_fp8_probe.py [out.json]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from sam3_lora_amd import _ffi, functional as Fn
from sam3_lora_amd.fp8 import Fp8Quantizer, Fp8Weight

DEV = "cuda:0"


def timed(f, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    M, fin, fout, rank, s = 8 * 5184, 1024, 4736, 16, 2.0
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    W = (torch.randn(fout, fin, device=DEV, generator=g) / 32).bfloat16()
    b = torch.randn(fout, device=DEV, generator=g).bfloat16()
    A = torch.randn(fin, rank, device=DEV, generator=g) / 32
    B = torch.randn(rank, fout, device=DEV, generator=g) * 0.05
    blob = Fn.pack_operands(A, B, 0, dtype=x.dtype)
    st = Fp8Weight(W)
    xq, sx = st.qx(x)
    qa = Fp8Quantizer(_ffi.FP8_E4M3)
    lib = _ffi.load()
    h0 = torch.addmm(b, x, W.t())
    qa(torch.nn.functional.gelu(h0.float()).bfloat16())           # calibrate the role "input of fc2"
    image = torch.empty(M, fout, dtype=torch.float8_e4m3fn, device=DEV)
    a = torch.empty(M, fout, dtype=torch.bfloat16, device=DEV)

    def slots():
        return (image, _ffi.FP8_E4M3) + qa.begin(x.device)

    def two_pass_fp8():
        h = torch._scaled_mm(xq, st.wq.t(), scale_a=sx, scale_b=st.scale, bias=b, out_dtype=torch.bfloat16)
        Fn.lora_fwd_(x, A, B, h, s, 0, save_t=False, packed=blob, gelu_out=a, q8=slots())

    def gemm_fp8_only():
        torch._scaled_mm(xq, st.wq.t(), scale_a=sx, scale_b=st.scale, bias=b, out_dtype=torch.bfloat16)

    def fused_fp8():
        Fn.lora_linear_fwd_q8_(x, xq, sx, st.wq, st.scale, b, A, B, s, 0, packed=blob, gelu=True, q8=slots())

    def fused_fp8_no_image():
        Fn.lora_linear_fwd_q8_(x, xq, sx, st.wq, st.scale, b, A, B, s, 0, packed=blob, gelu=True)

    def fused_bf16():
        Fn.lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=True)

    variants = [("two_pass_fp8 (scaled_mm + adapter pass with GELU + fp8 image)", two_pass_fp8), ("scaled_mm alone", gemm_fp8_only),
                ("fused_fp8 (one kernel, GELU + fp8 image)", fused_fp8), ("fused_fp8 without the image", fused_fp8_no_image),
                ("fused_bf16 (one kernel, GELU)", fused_bf16)]
    for _, f in variants:
        f()
    torch.cuda.synchronize()
    rounds = {}
    for r in range(5):
        for name, f in variants:
            f()
            rounds.setdefault(name, []).append(timed(f, 10))
    out = {"M": M, "in": fin, "out": fout, "rank": rank, "flop": 2.0 * M * fin * fout,
           "us": {k: {"median": float(np.median(v)), "min": float(np.min(v)), "all": [round(t, 1) for t in v]} for k, v in rounds.items()}}
    cap = 64
    lib.sam3_lora_prof_start(0xFFFFFFFF, cap)
    for _ in range(5):
        fused_fp8()
    us, stg, dm = (ctypes.c_float * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
    n = lib.sam3_lora_prof_stop(us, stg, dm, cap)
    agg = {}
    for i in range(n):
        agg.setdefault(f"stage{stg[i]}_dim{dm[i]}", []).append(us[i])
    out["insitu_fused_fp8"] = {k: round(float(np.median(v)), 2) for k, v in agg.items()}
    fz = out["insitu_fused_fp8"].get(f"stage{_ffi.STAGE_FUSED}_dim{fout}")
    if fz:
        out["fused_fp8_kernel_tflops"] = round(out["flop"] / fz / 1e6, 1)
        out["fused_fp8_kernel_frac_of_5PF_dense_fp8_peak"] = round(out["flop"] / fz / 1e6 / 5000.0, 3)
    txt = json.dumps(out)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
