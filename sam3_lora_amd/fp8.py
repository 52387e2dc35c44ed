"""
"fp8 frozen-W" mode of the frozen Linears around the adapters (BASELINE.json configs[4]: "mixed fp8 frozen-W / bf16 LoRA,
CDNA4 fp8 MFMA on base GEMMs").  A build-side extension -- the reference has no fp8 path.

  W            stored once as OCP e4m3 with one scale per tensor (it never changes), plus its transpose for the
               input-gradient GEMM;
  activations  bf16 -> e4m3 (forward) / e5m2 (incoming gradients) by the HIP quantiser (C-ABI ``sam3_fp8_quantize``,
               include/sam3_fp8_amd.h) with delayed scaling: a call scales with the amax its predecessor on the same
               tensor role observed -- but never with less than half the range the predecessor itself scaled with (the
               protocol's short memory: the range shrinks at most one octave per call, grows at once) -- and gathers the
               amax for its successor in the same pass;
  GEMMs        hipBLASLt fp8 MFMA kernels through ``torch._scaled_mm`` (measured on MI355X at M = 41,472: fc1 429 -> 268 us,
               fc2 317 -> 146 us, qkv 218 -> 126 us, proj 75 -> 48 us), bf16 out;
  LoRA branch  unchanged: bf16 activations through the adapter kernels, fp32 A / B.

Enabled per process with :func:`enable_fp8_frozen` (trainer: ``engine.fp8_frozen`` / ``--fp8-frozen``); only bf16 frozen
Linears whose widths are multiples of 16 take the fp8 route, everything else keeps its bf16 GEMM.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import Optional, Tuple

import torch

from . import _ffi

__all__ = ["enable_fp8_frozen", "fp8_enabled", "fp8_linear", "fp8_dx", "Fp8Quantizer", "Fp8Weight", "state_for", "eligible", "would_use", "producer_slots", "fp8_linear_q", "fp8_dx_q"]

_STATE = {"on": False}
_WEIGHTS = {}          # id(weight) -> (weakref to the weight, Fp8Weight); tensors compare element-wise, so not a dict key


def enable_fp8_frozen(on: bool = True) -> None:
    _STATE["on"] = bool(on)
    if not on:
        _WEIGHTS.clear()


def fp8_enabled() -> bool:
    return _STATE["on"]


class Fp8Quantizer:
    """One tensor role (e.g. "input of fc1"): delayed-scaling state + the quantise call.  The state -- two arrays of
    ``SAM3_FP8_AMAX_SLOTS`` amax slots (one 128-byte line each) used alternately (read by this call / gathered for the next) and the scale -- is also
    what the PRODUCING kernels take when they write the fp8 image themselves (:meth:`begin`)."""

    def __init__(self, fmt: int):
        self.fmt = fmt
        self.amax = None            # [2, SAM3_FP8_AMAX_FLOATS]: [read by this call, written for the next], alternating
        self.scale = None
        self.k = 0

    @property
    def torch_dtype(self):
        return torch.float8_e4m3fn if self.fmt == _ffi.FP8_E4M3 else torch.float8_e5m2

    def begin(self, device) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """(amax to scale with, zeroed amax to gather into, scale slot) for a kernel that produces the tensor and its fp8
        image in one pass -- or None on the very first use of this role (nothing to scale with yet: that call goes through
        :meth:`__call__`, which calibrates on the tensor itself)."""
        if self.amax is None or self.amax.device != device:
            return None
        cur, nxt = self.amax[self.k], self.amax[1 - self.k]
        nxt.zero_()
        self.k ^= 1
        return cur, nxt, self.scale

    def __call__(self, x2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        lib = _ffi.load()
        if self.amax is None or self.amax.device != x2.device:
            self.amax = torch.zeros(2, _ffi.FP8_AMAX_FLOATS, device=x2.device)
            self.scale = torch.ones(1, device=x2.device)        # what a call with nothing observed before it falls back to
            # first call: calibrate on the tensor itself (finite values only, as the kernels' running amax)
            self.amax[0, 0] = torch.nan_to_num(x2.detach().abs().float(), nan=0.0, posinf=0.0).max()
            self.k = 0
        cur, nxt = self.amax[self.k], self.amax[1 - self.k]
        nxt.zero_()
        fdt = self.torch_dtype
        out = torch.empty(x2.shape, dtype=fdt, device=x2.device)
        rc = lib.sam3_fp8_quantize(x2.data_ptr(), out.data_ptr(), cur.data_ptr(), nxt.data_ptr(), self.scale.data_ptr(),
                                   x2.numel(), 0 if x2.dtype == torch.bfloat16 else 1, self.fmt,
                                   ctypes.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sam3_fp8_quantize failed ({rc}): {(lib.sam3_fp8_last_error() or b'').decode()}")
        self.k ^= 1
        return out, self.scale


class Fp8Weight:
    """The frozen weight of one Linear in e4m3 (+ transpose), its scale, and the two activation quantisers."""

    def __init__(self, w: torch.Tensor):
        amax = w.detach().abs().max().float().clamp(min=2.0 ** -24)
        self.scale = (amax / 448.0).reshape(1)
        wq = (w.detach().float() / self.scale).clamp(-448.0, 448.0)
        self.wq = wq.to(torch.float8_e4m3fn)                                  # [N, K]
        self.wtq = wq.t().contiguous().to(torch.float8_e4m3fn)                # [K, N] for gy @ W
        self.qx = Fp8Quantizer(_ffi.FP8_E4M3)
        self.qg = Fp8Quantizer(_ffi.FP8_E5M2)
        self.stamp = (w.data_ptr(), w._version)


def eligible(x2: torch.Tensor, w: torch.Tensor) -> bool:
    return (_STATE["on"] and x2.is_cuda and x2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and not w.requires_grad
            and x2.dim() == 2 and x2.shape[0] > 0 and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0 and x2.is_contiguous()
            and not torch.is_autocast_enabled("cuda"))


def would_use(x: torch.Tensor, w: torch.Tensor) -> bool:
    """:func:`eligible` for an input of any rank before it is flattened to rows (the flattening makes it contiguous)."""
    return (_STATE["on"] and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and not w.requires_grad
            and x.numel() > 0 and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0 and not torch.is_autocast_enabled("cuda"))


def state_for(w: torch.Tensor) -> Fp8Weight:
    key = id(w)
    held = _WEIGHTS.get(key)
    if held is None or held[0]() is not w or held[1].stamp != (w.data_ptr(), w._version):
        st = Fp8Weight(w)
        _WEIGHTS[key] = (weakref.ref(w, lambda _, k=key: _WEIGHTS.pop(k, None)), st)
        return st
    return held[1]


def fp8_linear(x2: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """``F.linear(x2, w, b)`` with e4m3 operands on the fp8 MFMA GEMM, bf16 out."""
    st = state_for(w)
    xq, sx = st.qx(x2)
    return torch._scaled_mm(xq, st.wq.t(), scale_a=sx, scale_b=st.scale, bias=b, out_dtype=torch.bfloat16)


def fp8_linear_q(xq: torch.Tensor, sx: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """:func:`fp8_linear` on an input its producer already wrote as e4m3 (``xq`` with scale ``sx``)."""
    st = state_for(w)
    return torch._scaled_mm(xq, st.wq.t(), scale_a=sx, scale_b=st.scale, bias=b, out_dtype=torch.bfloat16)


def fp8_dx_q(gq: torch.Tensor, sg: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """:func:`fp8_dx` on a gradient its producer already wrote as e5m2."""
    st = state_for(w)
    return torch._scaled_mm(gq, st.wtq.t(), scale_a=sg, scale_b=st.scale, out_dtype=torch.bfloat16)


def producer_slots(w: torch.Tensor, role: str, rows: int, width: int, device):
    """For a kernel about to PRODUCE the [rows, width] bf16 input (role "x") or output gradient (role "g") of frozen weight
    ``w``: ``(image, fmt, amax_in, amax_out, scale)`` to hand to it so that it writes the fp8 image itself -- or None (mode
    off, widths the fp8 GEMM does not take, or first use of this role: the consumer then quantises separately)."""
    if os.environ.get("SAM3_FP8_SEPARATE_QUANT") == "1":      # A/B knob: every image by the stand-alone quantiser
        return None
    if not (_STATE["on"] and w.is_cuda and w.dtype == torch.bfloat16 and not w.requires_grad and rows > 0
            and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0 and not torch.is_autocast_enabled("cuda")):
        return None
    st = state_for(w)
    q = st.qx if role == "x" else st.qg
    slots = q.begin(device)
    if slots is None:
        return None
    image = torch.empty(rows, width, dtype=q.torch_dtype, device=device)
    return (image, q.fmt) + slots


def fp8_dx(gy2: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``gy2 @ w`` with the gradient in e5m2 and the weight in e4m3."""
    st = state_for(w)
    gq, sg = st.qg(gy2)
    return torch._scaled_mm(gq, st.wtq.t(), scale_a=sg, scale_b=st.scale, out_dtype=torch.bfloat16)
