"""Is k_fused_linear's L2 -> LDS fill bound per CU or by something the CUs share?  The fill-alone and matrix-pipe-alone probes
(SAM3_LORA_FUSED_PROBE = 1 / 2) and the kernel itself at 256 / 192 / 128 / 64 workgroups (one per CU): a per-CU bound scales the
time with 256 / WGS, a shared bound leaves it where it was."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam3_lora_amd import _ffi, functional as Fn   # noqa: E402
from tools.fused_linear_probe import timed   # noqa: E402

DEV = "cuda:0"
if __name__ == "__main__":
    M, fin, fout, rank, s = 41472, 1024, 4736, 16, 2.0
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    W = (torch.randn(fout, fin, device=DEV, generator=g) / 32).bfloat16()
    b = (torch.randn(fout, device=DEV, generator=g) * 0.1).bfloat16()
    A = (torch.rand(fin, rank, device=DEV, generator=g) - 0.5) / 2
    B = torch.randn(rank, fout, device=DEV, generator=g) * 0.05
    blob = Fn.pack_operands(A, B, 0)
    h = torch.empty(M, fout, device=DEV, dtype=torch.bfloat16)
    lib = _ffi.load()
    f = lambda: Fn.lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=False, y_out=h)
    res = {}
    for r in range(3):
        for wgs in [int(v) for v in os.environ.get("GRID_WGS", "256,192,128,64").split(",")]:
            for probe in [int(v) for v in os.environ.get("GRID_PROBES", "0,1,2").split(",")]:
                os.environ["SAM3_LORA_FUSED_WGS"] = str(wgs)
                os.environ["SAM3_LORA_FUSED_PROBE"] = str(probe)
                lib.sam3_lora_debug_reload_knobs()
                f()
                res.setdefault(f"wgs{wgs}_probe{probe}", []).append(timed(f, 6))
    out = {k: round(float(np.median(v)), 1) for k, v in res.items()}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(json.dumps(out, indent=1) + "\n")
