"""k_fused_linear as  tiles x (K steps x t_step + t_tile): the same M = 41,472 x N = 4736 output at in_features = 256 ... 4096
(K steps of 64 + the rank-r step), with and without GELU, hipBLASLt beside it.  The slope over K is the time of one K step
(fill + matrix pipe), the intercept the per-tile cost that does not depend on K (epilogue, first fill, tile switch).
Writes one JSON object to stdout (and to argv[1] if given)."""
import json
import os
import sys

import numpy as np
import torch

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
    M, fout, rank, s = 41472, int(os.environ.get("PROBE_N", 4736)), 16, 2.0
    g = torch.Generator(device=DEV).manual_seed(0)
    lib = _ffi.load()
    h = torch.empty(M, fout, device=DEV, dtype=torch.bfloat16)
    a = torch.empty_like(h)
    res = {}
    fins = (256, 512, 1024, 2048, 4096)
    cases = {}
    for fin in fins:
        x = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
        W = (torch.randn(fout, fin, device=DEV, generator=g) / 32).bfloat16()
        b = (torch.randn(fout, device=DEV, generator=g) * 0.1).bfloat16()
        A = (torch.rand(fin, rank, device=DEV, generator=g) - 0.5) / 2
        B = torch.randn(rank, fout, device=DEV, generator=g) * 0.05
        blob = Fn.pack_operands(A, B, 0)
        cases[fin] = (x, W, b, A, B, blob)
    for r in range(4):
        for fin in fins:
            x, W, b, A, B, blob = cases[fin]
            fs = {"gelu": lambda: Fn.lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=True, y_out=h, gelu_out=a),
                  "noact": lambda: Fn.lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=False, y_out=h),
                  "hipblaslt": lambda: torch.addmm(b, x, W.t(), out=h)}
            for name, f in fs.items():
                f()
                res.setdefault(name, {}).setdefault(fin, []).append(timed(f, 8))
    out = {"M": M, "out": fout, "rank": rank, "site_us_median": {k: {fin: round(float(np.median(v)), 1) for fin, v in d.items()} for k, d in res.items()}}
    tiles_per_cu = -(-M // 256) * (fout / 256) / 256
    out["tiles_per_cu"] = round(tiles_per_cu, 2)
    for name, d in res.items():
        ks = np.array([fin / 64 for fin in fins])
        ts = np.array([float(np.median(d[fin])) for fin in fins])
        slope, icpt = np.polyfit(ks, ts, 1)
        out[f"{name}_fit"] = {"us_per_K64_step_all_tiles": round(float(slope), 2), "intercept_us": round(float(icpt), 1),
                              "us_per_step_per_tile": round(float(slope / tiles_per_cu), 3), "us_per_tile_fixed": round(float(icpt / tiles_per_cu), 2)}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
