"""
Seeded input generator shared by ``make_golden.py`` (which runs the reference on these
inputs, in the build container) and by the tests (which regenerate the same inputs and
compare against the stored reference outputs).  Inputs are NOT stored in the fixtures.

All tensors are bf16-representable fp32 so that the fp32 reference and a bf16 device
path see bit-identical operands.
"""
from __future__ import annotations

import zlib

import numpy as np

LAYOUT_ROOT = 0
LAYOUT_PACKAGE = 1

# name -> (layout, M, in, out, rank, alpha)
# fc1/fc2 are the SAM3 ViT MLP shapes the root injector adapts; the others
# are a DETR-FFN-like shape at the remaining ranks BASELINE.json's configs name.
CASES = {
    "root_fc1_r16": (LAYOUT_ROOT, 19, 1024, 4736, 16, 32),
    "root_fc2_r16": (LAYOUT_ROOT, 19, 4736, 1024, 16, 32),
    "pkg_fc1_r16": (LAYOUT_PACKAGE, 19, 1024, 4736, 16, 32.0),
    "pkg_fc2_r16": (LAYOUT_PACKAGE, 19, 4736, 1024, 16, 32.0),
    "root_ffn_r4": (LAYOUT_ROOT, 37, 256, 2048, 4, 8),
    "root_ffn_r8": (LAYOUT_ROOT, 37, 256, 2048, 8, 16),
    "root_ffn_r32": (LAYOUT_ROOT, 37, 2048, 256, 32, 64),
    "pkg_ffn_r4": (LAYOUT_PACKAGE, 37, 256, 2048, 4, 1.0),
    "pkg_ffn_r8": (LAYOUT_PACKAGE, 37, 2048, 256, 8, 16.0),
    "pkg_ffn_r32": (LAYOUT_PACKAGE, 37, 256, 2048, 32, 64.0),
    # batched leading dims (x is [B, T, in]) and a single row
    "root_3d_r16": (LAYOUT_ROOT, (2, 5), 256, 512, 16, 32),
    "pkg_1row_r16": (LAYOUT_PACKAGE, 1, 256, 512, 16, 8.0),
}


def bf16_round(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def make_case(name: str) -> dict:
    layout, M, fin, fout, rank, alpha = CASES[name]
    lead = M if isinstance(M, tuple) else (M,)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    n = lambda *s: rng.standard_normal(s, dtype=np.float32)
    x = bf16_round(n(*lead, fin))
    W = bf16_round(n(fout, fin) / np.float32(np.sqrt(fin)))
    b = bf16_round(n(fout) * np.float32(0.1))
    gy = bf16_round(n(*lead, fout))
    if layout == LAYOUT_ROOT:
        A = bf16_round(rng.uniform(-1, 1, (fin, rank)).astype(np.float32) / np.float32(np.sqrt(rank)))
        B = bf16_round(n(rank, fout) * np.float32(0.05))
    else:
        A = bf16_round(rng.uniform(-1, 1, (rank, fin)).astype(np.float32) / np.float32(np.sqrt(fin)))
        B = bf16_round(n(fout, rank) * np.float32(0.05))
    return dict(name=name, layout=layout, x=x, W=W, b=b, gy=gy, A=A, B=B,
                rank=rank, alpha=alpha, scaling=alpha / rank)
