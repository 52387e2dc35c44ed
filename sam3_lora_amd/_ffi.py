"""
ctypes binding of the C-ABI in include/sam3_lora_amd.h.

There is deliberately NO fallback: if the shared library is missing or does not export the
expected symbols, importing/using the LoRA forward raises.  (The library is built in-tree by
``__graft_entry__.build()`` / ``python -m sam3_lora_amd.build``.)
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

from .build import LIB_PATH

LAYOUT_ROOT = 0
LAYOUT_PACKAGE = 1
DT_BF16 = 0
DT_F32 = 1
ABI_VERSION = 6
MAX_RANK = 1024

EXPORTS = (
    "sam3_lora_abi_version", "sam3_lora_last_error", "sam3_lora_saved_t_bytes",
    "sam3_lora_fwd_workspace_bytes", "sam3_lora_bwd_workspace_bytes",
    "sam3_lora_fwd", "sam3_lora_bwd", "sam3_lora_merge", "sam3_lora_debug_set_stages", "sam3_lora_debug_reload_knobs",
    "sam3_lora_prof_start", "sam3_lora_prof_stop",
    "sam3_lora_packed_bytes", "sam3_lora_pack", "sam3_lora_pack_many", "sam3_lora_fwd_act", "sam3_lora_bwd_act",
    "sam3_lora_fwd_act_q8", "sam3_lora_bwd_act_q8", "sam3_lora_bwd_act_recomputes_input",
    "sam3_lora_linear_fwd", "sam3_lora_linear_fwd_supported", "sam3_lora_linear_fwd_workspace_bytes", "sam3_lora_linear_fwd_q8",
    "sam3_lora_linear_dgrad_act",
)
ACT_NONE, ACT_GELU = 0, 1
PREPACKED = 0x100
VIT_EXPORTS = ("sam3_vit_qkv_rope_fwd", "sam3_vit_qkv_rope_bwd", "sam3_vit_qkv_rope_win_fwd",
               "sam3_vit_qkv_rope_win_bwd", "sam3_vit_win_residual", "sam3_vit_layernorm_fwd",
               "sam3_vit_layernorm_bwd", "sam3_vit_layernorm_bwd_add", "sam3_vit_layernorm_fwd_q8")      # include/sam3_vit_amd.h
LOSS_EXPORTS = ("sam3_loss_last_error", "sam3_mask_loss_workspace_bytes", "sam3_mask_loss_fwd",
                "sam3_mask_loss_bwd", "sam3_box_pair_fwd", "sam3_box_pair_bwd")                                          # include/sam3_loss_amd.h
FP8_EXPORTS = ("sam3_fp8_last_error", "sam3_fp8_quantize")                 # include/sam3_fp8_amd.h
SEG_EXPORTS = ("sam3_seg_last_error", "sam3_gn_nhwc_supported", "sam3_gn_nhwc_workspace_bytes", "sam3_gn_nhwc_fwd",
               "sam3_gn_nhwc_bwd", "sam3_rpb_bias_fwd")                                             # include/sam3_seg_amd.h
FP8_E4M3, FP8_E5M2 = 0, 1
FP8_AMAX_SLOTS = 64            # SAM3_FP8_AMAX_SLOTS
FP8_AMAX_STRIDE = 32           # SAM3_FP8_AMAX_STRIDE: floats between two slots (one 128-byte line per slot)
FP8_AMAX_FLOATS = FP8_AMAX_SLOTS * FP8_AMAX_STRIDE
STAGE_PACK, STAGE_T1, STAGE_T2, STAGE_T3_GB, STAGE_T3_GA, STAGE_REDUCE, STAGE_ALL = 1, 2, 4, 8, 16, 32, 0xFFFFFFFF
STAGE_GT_REDUCE, STAGE_FUSED, STAGE_T3W, STAGE_XGX = 64, 128, 256, 512

_lib = None
_lock = threading.Lock()


class LoRAKernelError(RuntimeError):
    pass


def _declare(lib):
    lib.sam3_lora_abi_version.restype = c_int
    lib.sam3_lora_abi_version.argtypes = []
    lib.sam3_lora_last_error.restype = c_char_p
    lib.sam3_lora_last_error.argtypes = []
    lib.sam3_lora_saved_t_bytes.restype = c_size_t
    lib.sam3_lora_saved_t_bytes.argtypes = [c_int64, c_int, c_int]
    for f in (lib.sam3_lora_fwd_workspace_bytes, lib.sam3_lora_bwd_workspace_bytes):
        f.restype = c_size_t
        f.argtypes = [c_int64, c_int, c_int, c_int, c_int]
    lib.sam3_lora_packed_bytes.restype = c_size_t
    lib.sam3_lora_packed_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.sam3_lora_pack.restype = c_int
    lib.sam3_lora_pack.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.sam3_lora_pack_many.restype = c_int
    lib.sam3_lora_pack_many.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.sam3_lora_fwd.restype = c_int
    lib.sam3_lora_fwd.argtypes = [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,      # x, A, B, y_inout, tT_out
        c_int64, c_int, c_int, c_int,                          # M, in, out, rank
        c_int64, c_int64, c_int, c_float,                      # ldx, ldy, layout, scaling
        c_float, c_uint64, c_uint64, c_int,                    # drop_p, seed, offset, dtype
        c_void_p, c_size_t, c_void_p,                          # workspace, bytes, stream
    ]
    lib.sam3_lora_bwd.restype = c_int
    lib.sam3_lora_bwd.argtypes = [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,      # gy, x, tT_saved, A, B
        c_void_p, c_void_p, c_void_p,                          # gx_inout, gA_accum, gB_accum
        c_int64, c_int, c_int, c_int,                          # M, in, out, rank
        c_int64, c_int64, c_int64, c_int, c_float,             # ldgy, ldx, ldgx, layout, scaling
        c_float, c_uint64, c_uint64, c_int, c_int,             # drop_p, seed, offset, dtype, accumulate
        c_void_p, c_size_t, c_void_p,                          # workspace, bytes, stream
    ]
    lib.sam3_lora_fwd_act.restype = c_int
    lib.sam3_lora_fwd_act.argtypes = list(lib.sam3_lora_fwd.argtypes) + [c_int, c_void_p, c_int64]
    lib.sam3_lora_bwd_act.restype = c_int
    lib.sam3_lora_bwd_act.argtypes = list(lib.sam3_lora_bwd.argtypes) + [c_int, c_void_p, c_int64]
    q8_tail = [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]     # q8_out, ldq, fmt, amax_in, amax_out, scale_out
    lib.sam3_lora_fwd_act_q8.restype = c_int
    lib.sam3_lora_fwd_act_q8.argtypes = list(lib.sam3_lora_fwd_act.argtypes) + q8_tail
    lib.sam3_lora_bwd_act_q8.restype = c_int
    lib.sam3_lora_bwd_act_q8.argtypes = list(lib.sam3_lora_bwd_act.argtypes) + q8_tail
    lib.sam3_lora_linear_fwd_supported.restype = c_int
    lib.sam3_lora_linear_fwd_supported.argtypes = [c_int, c_int, c_int, c_int]
    lib.sam3_lora_linear_fwd_workspace_bytes.restype = c_size_t
    lib.sam3_lora_linear_fwd_workspace_bytes.argtypes = [c_int64, c_int, c_int, c_int, c_int]
    lib.sam3_lora_linear_fwd.restype = c_int
    lib.sam3_lora_linear_fwd.argtypes = [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,   # x, W, bias, A, B, y_out, tT_out
        c_int64, c_int, c_int, c_int,                          # M, in, out, rank
        c_int64, c_int64, c_int64, c_int, c_float,             # ldx, ldw, ldy, layout, scaling
        c_float, c_uint64, c_uint64, c_int,                    # drop_p, seed, offset, dtype
        c_void_p, c_size_t, c_void_p,                          # workspace, bytes, stream
        c_int, c_void_p, c_int64,                              # act, act_out, ldact
    ]
    lib.sam3_lora_linear_fwd_q8.restype = c_int
    lib.sam3_lora_linear_fwd_q8.argtypes = [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,     # x, x_q8, ldxq, scale_x, w_q8, ldwq, scale_w
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,       # bias, A, B, y_out, tT_out
        c_int64, c_int, c_int, c_int,                          # M, in, out, rank
        c_int64, c_int64, c_int, c_float,                      # ldx, ldy, layout, scaling
        c_float, c_uint64, c_uint64, c_int,                    # drop_p, seed, offset, dtype
        c_void_p, c_size_t, c_void_p,                          # workspace, bytes, stream
        c_int, c_void_p, c_int64,                              # act, act_out, ldact
    ] + q8_tail
    lib.sam3_lora_linear_dgrad_act.restype = c_int
    lib.sam3_lora_linear_dgrad_act.argtypes = [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,      # gy, Wt, A, B, gx_out
        c_int64, c_int, c_int, c_int,                          # M, in, out, rank
        c_int64, c_int64, c_int64, c_int, c_float, c_int,      # ldgy, ldwt, ldgx, layout, scaling, dtype
        c_void_p, c_size_t, c_void_p,                          # workspace, bytes, stream
        c_int, c_void_p, c_int64,                              # act, pre_act, ldpre
    ]
    lib.sam3_lora_bwd_act_recomputes_input.restype = c_int
    lib.sam3_lora_bwd_act_recomputes_input.argtypes = [c_int, c_int, ctypes.c_float]
    lib.sam3_lora_debug_reload_knobs.restype = None
    lib.sam3_lora_debug_reload_knobs.argtypes = []
    lib.sam3_lora_debug_set_stages.restype = ctypes.c_uint
    lib.sam3_lora_debug_set_stages.argtypes = [ctypes.c_uint]
    lib.sam3_lora_prof_start.restype = c_int
    lib.sam3_lora_prof_start.argtypes = [ctypes.c_uint, c_int]
    lib.sam3_lora_prof_stop.restype = c_int
    lib.sam3_lora_prof_stop.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
    for f in (lib.sam3_vit_qkv_rope_fwd, lib.sam3_vit_qkv_rope_bwd):
        f.restype = c_int
    lib.sam3_vit_qkv_rope_fwd.argtypes = [c_void_p] * 6 + [c_int64, c_int, c_int, c_int, c_int, c_void_p]
    lib.sam3_vit_qkv_rope_bwd.argtypes = [c_void_p] * 3 + [c_int64] * 3 + [c_void_p] * 3 + [c_int64, c_int, c_int, c_int,
                                                                                      c_int, c_void_p]
    for f in (lib.sam3_vit_qkv_rope_win_fwd, lib.sam3_vit_qkv_rope_win_bwd, lib.sam3_vit_win_residual):
        f.restype = c_int
    lib.sam3_vit_qkv_rope_win_fwd.argtypes = [c_void_p] * 6 + [c_int64] + [c_int] * 7 + [c_void_p]
    lib.sam3_vit_qkv_rope_win_bwd.argtypes = ([c_void_p] * 3 + [c_int64] * 3 + [c_void_p] * 3 + [c_int64] + [c_int] * 7
                                              + [c_void_p])
    lib.sam3_vit_win_residual.argtypes = [c_void_p] * 4 + [c_int64] + [c_int] * 6 + [c_void_p]
    lib.sam3_vit_layernorm_fwd.restype = c_int
    lib.sam3_vit_layernorm_fwd.argtypes = [c_void_p] * 6 + [c_int64, c_int, c_float, c_int, c_void_p]
    lib.sam3_vit_layernorm_fwd_q8.restype = c_int
    lib.sam3_vit_layernorm_fwd_q8.argtypes = [c_void_p] * 6 + [c_int64, c_int, c_float, c_int, c_void_p, c_int64, c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p]
    lib.sam3_vit_layernorm_bwd.restype = c_int
    lib.sam3_vit_layernorm_bwd.argtypes = [c_void_p] * 6 + [c_int64, c_int, c_int, c_void_p]
    lib.sam3_vit_layernorm_bwd_add.restype = c_int
    lib.sam3_vit_layernorm_bwd_add.argtypes = [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p]
    lib.sam3_loss_last_error.restype = c_char_p
    lib.sam3_loss_last_error.argtypes = []
    lib.sam3_mask_loss_workspace_bytes.restype = c_size_t
    lib.sam3_mask_loss_workspace_bytes.argtypes = [c_int, c_int, c_int]
    lib.sam3_mask_loss_fwd.restype = c_int
    lib.sam3_mask_loss_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                       c_int, c_void_p, c_size_t, c_void_p]
    lib.sam3_mask_loss_bwd.restype = c_int
    lib.sam3_mask_loss_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_float, c_int, c_int, c_void_p]
    lib.sam3_box_pair_fwd.restype = c_int
    lib.sam3_box_pair_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    lib.sam3_box_pair_bwd.restype = c_int
    lib.sam3_box_pair_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    lib.sam3_fp8_last_error.restype = c_char_p
    lib.sam3_fp8_last_error.argtypes = []
    lib.sam3_fp8_quantize.restype = c_int
    lib.sam3_fp8_quantize.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]
    lib.sam3_seg_last_error.restype = c_char_p
    lib.sam3_seg_last_error.argtypes = []
    lib.sam3_gn_nhwc_supported.restype = c_int
    lib.sam3_gn_nhwc_supported.argtypes = [c_int, c_int, c_int]
    lib.sam3_gn_nhwc_workspace_bytes.restype = c_size_t
    lib.sam3_gn_nhwc_workspace_bytes.argtypes = [c_int, c_int64, c_int, c_int]
    lib.sam3_gn_nhwc_fwd.restype = c_int
    lib.sam3_gn_nhwc_fwd.argtypes = [c_void_p] * 5 + [c_int, c_int64, c_int, c_int, c_float, c_int, c_int, c_void_p,
                                     c_size_t, c_void_p]
    lib.sam3_gn_nhwc_bwd.restype = c_int
    lib.sam3_gn_nhwc_bwd.argtypes = [c_void_p] * 6 + [c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                     c_void_p]
    lib.sam3_rpb_bias_fwd.restype = c_int
    lib.sam3_rpb_bias_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]
    lib.sam3_lora_merge.restype = c_int
    lib.sam3_lora_merge.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_float, c_void_p]


def load(path: str | None = None):
    """Load (once) and return the ctypes handle.  Raises LoRAKernelError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        p = path or os.environ.get("SAM3_LORA_AMD_LIB") or LIB_PATH
        if p == LIB_PATH:
            # a fresh clone has no binary (git-ignored) and an edited kernel source makes it stale: build in-tree
            # when hipcc is there; without hipcc an existing binary is used as it is, a missing one is an error
            from . import build as _build
            if _build.needs_build():
                try:
                    _build.build_library()
                except Exception as e:
                    if not os.path.exists(p):
                        raise LoRAKernelError(f"sam3_lora_amd: {p} is missing and could not be built: {e}") from e
        if not os.path.exists(p):
            raise LoRAKernelError(
                f"sam3_lora_amd: HIP library not found at {p}. Build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                f"There is no CPU/PyTorch fallback for the LoRA path.")
        try:
            lib = ctypes.CDLL(p)
        except OSError as e:  # missing libamdhip64 etc.
            raise LoRAKernelError(f"sam3_lora_amd: cannot load {p}: {e}") from e
        missing = [s for s in EXPORTS + VIT_EXPORTS + LOSS_EXPORTS + FP8_EXPORTS + SEG_EXPORTS if not hasattr(lib, s)]
        if missing:
            raise LoRAKernelError(f"sam3_lora_amd: {p} lacks symbols {missing}")
        _declare(lib)
        v = lib.sam3_lora_abi_version()
        if v != ABI_VERSION:
            raise LoRAKernelError(f"sam3_lora_amd: ABI version {v} != expected {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def last_error() -> str:
    return (load().sam3_lora_last_error() or b"").decode()


def check(rc: int, what: str):
    if rc != 0:
        raise LoRAKernelError(f"{what} failed (code {rc}): {last_error()}")
