"""SURVEY 8(f)-1 measured: the fc1 -> GELU site at BASELINE configs[1] (M = 41,472, 1024 -> 4736, r = 16, bf16) as
  (a) hipBLASLt GEMM + sam3_lora_fwd_act (k_t1 + k_t2<GELU>)      -- the two-pass form
  (b) sam3_lora_linear_fwd (k_t1 + k_wext + k_fused_linear)       -- the adapter inside the GEMM
interleaved rounds in one process, HIP-event timed on the current stream; the in-situ profiler splits (b) into its kernels.
Writes one JSON object to stdout (and to argv[1] if given)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam3_lora_amd import _ffi, functional as Fn   # noqa: E402

DEV = "cuda:0"


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    M, fin, fout, rank, s = int(os.environ.get("PROBE_M", 41472)), 1024, 4736, 16, 2.0
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    W = (torch.randn(fout, fin, device=DEV, generator=g) / 32).bfloat16()
    b = (torch.randn(fout, device=DEV, generator=g) * 0.1).bfloat16()
    A = (torch.rand(fin, rank, device=DEV, generator=g) - 0.5) / 2
    B = torch.randn(rank, fout, device=DEV, generator=g) * 0.05
    blob = Fn.pack_operands(A, B, 0)
    h = torch.empty(M, fout, device=DEV, dtype=torch.bfloat16)
    a = torch.empty_like(h)
    lib = _ffi.load()

    def two_pass():
        torch.addmm(b, x, W.t(), out=h)
        Fn.lora_fwd_(x, A, B, h, s, 0, save_t=False, packed=blob, gelu_out=a)

    def gemm_only():
        torch.addmm(b, x, W.t(), out=h)

    def fused():
        Fn.lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=True, y_out=h, gelu_out=a)

    def fused_noact():
        Fn.lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=False, y_out=h)

    out = {"M": M, "in": fin, "out": fout, "rank": rank, "flop": 2.0 * M * fin * fout}
    for f in (two_pass, gemm_only, fused, fused_noact):
        f()
    torch.cuda.synchronize()
    rounds = {}
    variants = [("two_pass", two_pass, {}), ("gemm_only", gemm_only, {}),
                ("fused_big", fused, {}), ("fused_noact_big", fused_noact, {}),
                ("fused_big_last_column_as_full_tiles", fused, {"SAM3_LORA_FUSED_HALF": "0"}),
                # what bounds the kernel (fl::k_fused_linear's PROBE; outputs are garbage in these two):
                ("PROBE_fill_only_gelu", fused, {"SAM3_LORA_FUSED_PROBE": "1"}), ("PROBE_fill_only_noact", fused_noact, {"SAM3_LORA_FUSED_PROBE": "1"}),
                ("PROBE_mfma_only_gelu", fused, {"SAM3_LORA_FUSED_PROBE": "2"}), ("PROBE_mfma_only_noact", fused_noact, {"SAM3_LORA_FUSED_PROBE": "2"}),
                ("PROBE_no_global_stores_gelu", fused, {"SAM3_LORA_FUSED_PROBE": "3"}), ("PROBE_no_global_stores_noact", fused_noact, {"SAM3_LORA_FUSED_PROBE": "3"}),
                ("PROBE_no_gelu_arithmetic", fused, {"SAM3_LORA_FUSED_PROBE": "4"})]
    for r in range(5):
        for name, f, env in variants:
            for k in ("SAM3_LORA_FUSED_WGS", "SAM3_LORA_FUSED_HALF", "SAM3_LORA_FUSED_PROBE"):
                os.environ.pop(k, None)
            os.environ.update(env)
            lib.sam3_lora_debug_reload_knobs()
            f()
            rounds.setdefault(name, []).append(timed(f, 10))
    for k in ("SAM3_LORA_FUSED_WGS", "SAM3_LORA_FUSED_HALF", "SAM3_LORA_FUSED_PROBE"):
        os.environ.pop(k, None)
    lib.sam3_lora_debug_reload_knobs()
    out["us"] = {k: {"median": float(np.median(v)), "min": float(np.min(v)), "all": [round(t, 1) for t in v]} for k, v in rounds.items()}
    # in-situ split of the fused call and of the two-pass adapter call
    cap = 64
    for name, f in (("fused", fused), ("two_pass", two_pass)):
        lib.sam3_lora_prof_start(0xFFFFFFFF, cap)
        for _ in range(5):
            f()
        us = (ctypes.c_float * cap)()
        st = (ctypes.c_int * cap)()
        dm = (ctypes.c_int * cap)()
        n = lib.sam3_lora_prof_stop(us, st, dm, cap)
        agg = {}
        for i in range(n):
            agg.setdefault(f"stage{st[i]}_dim{dm[i]}", []).append(us[i])
        out[f"insitu_{name}"] = {k: round(float(np.median(v)), 2) for k, v in agg.items()}
    fz = out["insitu_fused"].get(f"stage{_ffi.STAGE_FUSED}_dim{fout}")
    if fz:
        out["fused_kernel_tflops"] = round(out["flop"] / fz / 1e6, 1)
        out["fused_kernel_write_GBps"] = round(2 * M * fout * 2 / fz / 1e3, 1)
    gm = out["us"]["gemm_only"]["median"]
    out["hipblaslt_tflops"] = round(out["flop"] / gm / 1e6, 1)
    # correctness spot check in the same run
    rows = torch.randint(0, M, (512,), generator=torch.Generator().manual_seed(1)).to(DEV)
    fused()
    xd = x[rows].double()
    want = xd @ W.double().t() + b.double() + s * ((xd @ A.double()) @ B.double())
    out["max_rel_err_vs_fp64"] = float(((h[rows].double() - want).abs() / (want.abs() + 1e-2)).max())
    txt = json.dumps(out)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
