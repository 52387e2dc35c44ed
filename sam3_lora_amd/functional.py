"""
Host-side glue between torch tensors and the C-ABI (include/sam3_lora_amd.h).

``lora_linear`` is the single differentiable entry point both reference-compatible module
families use (``sam3_lora_amd.lora_layers.LoRALinear`` <- lora_layers.py:58-91 and
``sam3_lora_amd.lora.LinearWithLoRA`` <- sam3_lora/lora/lora_layer.py:91-158):

    y = F.linear(x, W, b)  +  scaling * (drop(x) @ A_c) @ B_c

The frozen GEMMs (F.linear forward, ``gy @ W`` backward) run on PyTorch-ROCm (hipBLASLt);
everything involving A and B runs in the hand-written HIP kernels, in place on the GEMM
outputs.  torch is plumbing here: device memory, the current HIP stream, autograd wiring.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _ffi
from ._ffi import DT_BF16, DT_F32, LAYOUT_PACKAGE, LAYOUT_ROOT, PREPACKED, LoRAKernelError

__all__ = ["lora_linear", "lora_fwd_", "lora_bwd_", "lora_linear_fwd_", "linear_fwd_supported", "set_fused_linear", "fused_linear_enabled", "merge_weight", "pack_operands", "PackedOperands", "lora_mlp_gelu", "TransposedCopy", "frozen_linear", "LAYOUT_ROOT", "LAYOUT_PACKAGE",
           "enable_direct_grad_accumulation", "direct_grad_accumulation", "pack_operands_many", "repack_adapters"]

_ws_lock = threading.Lock()
_workspaces = {}  # (device index, stream handle) -> uint8 tensor


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Scratch buffer reused by consecutive calls on one stream (stream order makes that safe)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    with _ws_lock:
        buf = _workspaces.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            _workspaces[key] = buf
    return buf


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    raise LoRAKernelError(f"sam3_lora_amd: activations must be bfloat16 or float32, got {t.dtype}")


def _require_cuda(*ts: torch.Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise LoRAKernelError(
                "sam3_lora_amd: the LoRA path runs only on an AMD GPU through the HIP kernels "
                f"(got a {t.device} tensor); there is no CPU fallback.")


def _rows(t: torch.Tensor) -> torch.Tensor:
    """[..., C] -> [M, C] with unit column stride and 16-byte aligned rows (copies only if needed)."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or (t2.stride(0) * t2.element_size()) % 16 or t2.data_ptr() % 16:
        t2 = t2.contiguous()
    return t2


def _master(p: torch.Tensor) -> torch.Tensor:
    p = p.detach()
    if p.dtype != torch.float32 or not p.is_contiguous():
        p = p.float().contiguous()
    return p


def _rank_of(A: torch.Tensor, layout: int) -> int:
    return A.shape[1] if layout == LAYOUT_ROOT else A.shape[0]


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _pad_cols(t2: torch.Tensor, width: int) -> torch.Tensor:
    if t2.shape[1] == width:
        return t2
    out = t2.new_zeros(t2.shape[0], width)
    out[:, :t2.shape[1]] = t2
    return out


def _pad_masters(A: torch.Tensor, B: torch.Tensor, layout: int, fin: int, fout: int, fin_p: int, fout_p: int):
    """Zero-padded fp32 copies of A/B (caller layout) for feature widths rounded up to a multiple of 8."""
    r = _rank_of(A, layout)
    if layout == LAYOUT_ROOT:
        Ap, Bp = A.new_zeros(fin_p, r), B.new_zeros(r, fout_p)
        Ap[:fin], Bp[:, :fout] = A, B
    else:
        Ap, Bp = A.new_zeros(r, fin_p), B.new_zeros(fout_p, r)
        Ap[:, :fin], Bp[:fout] = A, B
    return Ap, Bp


def saved_t_like(M: int, rank: int, device, dt: int = DT_BF16) -> torch.Tensor:
    n = _ffi.load().sam3_lora_saved_t_bytes(M, rank, dt)
    return torch.empty(n, dtype=torch.uint8, device=device)


def _dt_of(dtype) -> int:
    """Activation dtype (torch dtype or DT_* code) -> DT_* code."""
    if isinstance(dtype, int):
        return dtype
    if dtype == torch.bfloat16:
        return DT_BF16
    if dtype == torch.float32:
        return DT_F32
    raise LoRAKernelError(f"sam3_lora_amd: activations must be bfloat16 or float32, got {dtype}")


def pack_operands(A: torch.Tensor, B: torch.Tensor, layout: int, out: Optional[torch.Tensor] = None,
                  dtype=torch.bfloat16) -> torch.Tensor:
    """Operand images of the current A/B for ``lora_fwd_/lora_bwd_(..., packed=blob)`` (sam3_lora_pack): bf16 images for
    bf16 activations, fp32 images for the exact-fp32 path -- ``dtype`` is the ACTIVATION dtype the blob will be used with."""
    lib = _ffi.load()
    _require_cuda(A, B)
    dt = _dt_of(dtype)
    rank = _rank_of(A, layout)
    fin = A.shape[0] if layout == LAYOUT_ROOT else A.shape[1]
    fout = B.shape[1] if layout == LAYOUT_ROOT else B.shape[0]
    n = lib.sam3_lora_packed_bytes(fin, fout, rank, dt)
    if n == 0:
        raise LoRAKernelError(f"sam3_lora_packed_bytes: {_ffi.last_error()}")
    if out is None or out.numel() != n or out.device != A.device:
        out = torch.empty(n, dtype=torch.uint8, device=A.device)
    rc = lib.sam3_lora_pack(A.data_ptr(), B.data_ptr(), out.data_ptr(), fin, fout, rank, layout, dt,
                            ctypes.c_void_p(torch.cuda.current_stream(A.device).cuda_stream))
    _ffi.check(rc, "sam3_lora_pack")
    return out


def pack_operands_many(pairs, layout: int, dtype=torch.bfloat16, outs=None):
    """:func:`pack_operands` for a list of ``(A, B)`` pairs of one layout with ONE C-ABI call (``sam3_lora_pack_many``:
    the images of 16 adapters per launch).  ``outs``: blobs to overwrite (entries of the right size are reused)."""
    lib = _ffi.load()
    dt = _dt_of(dtype)
    n = len(pairs)
    if n == 0:
        return []
    outs = list(outs) if outs is not None else [None] * n
    fins, fouts, ranks = [], [], []
    for i, (A, B) in enumerate(pairs):
        _require_cuda(A, B)
        rank = _rank_of(A, layout)
        fin = A.shape[0] if layout == LAYOUT_ROOT else A.shape[1]
        fout = B.shape[1] if layout == LAYOUT_ROOT else B.shape[0]
        nb = lib.sam3_lora_packed_bytes(fin, fout, rank, dt)
        if nb == 0:
            raise LoRAKernelError(f"sam3_lora_packed_bytes: {_ffi.last_error()}")
        if outs[i] is None or outs[i].numel() != nb or outs[i].device != A.device:
            outs[i] = torch.empty(nb, dtype=torch.uint8, device=A.device)
        fins.append(fin), fouts.append(fout), ranks.append(rank)
    ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    ints = lambda v: (ctypes.c_int * n)(*v)
    dev = pairs[0][0].device
    rc = lib.sam3_lora_pack_many(n, ptrs([a for a, _ in pairs]), ptrs([b for _, b in pairs]), ptrs(outs), ints(fins),
                                 ints(fouts), ints(ranks), int(layout), dt,
                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _ffi.check(rc, "sam3_lora_pack_many")
    return outs


def repack_adapters(model: torch.nn.Module) -> int:
    """Refresh the cached operand images of every adapter of ``model`` in one batched call per (layout, dtype) -- call
    it right after ``optimizer.step()``: every A / B just changed, and the lazy per-layer refresh would otherwise cost
    one small launch per adapter in the next forward (64 for the SAM3 trunk).  Blobs are overwritten in place, so call
    it between steps (no autograd graph of the previous step alive).  Returns the number of adapters packed."""
    groups = {}
    for m in model.modules():
        lora = getattr(m, "lora", None)
        base = getattr(m, "original_layer", None) or getattr(m, "linear", None)
        cache = getattr(lora, "_packed", None)
        if cache is None or base is None or not isinstance(cache, PackedOperands):
            continue
        A, B = lora.lora_A, lora.lora_B
        if not (A.is_cuda and _is_master(A) and _is_master(B)) or base.weight.dtype not in (torch.bfloat16, torch.float32):
            continue
        layout = LAYOUT_ROOT if A.shape[1] == getattr(lora, "rank", -1) and A.shape[0] == base.in_features else LAYOUT_PACKAGE
        if base.in_features % 8 or base.out_features % 8:
            continue
        groups.setdefault((layout, _dt_of(base.weight.dtype), A.device), []).append((cache, A, B))
    total = 0
    for (layout, dt, _), items in groups.items():
        olds = [c._held.get(dt, (None, None))[1] for c, _, _ in items]
        blobs = pack_operands_many([(_master(A), _master(B)) for _, A, B in items], layout, dtype=dt, outs=olds)
        for (c, A, B), blob in zip(items, blobs):
            c._held[dt] = ((A.data_ptr(), A._version, B.data_ptr(), B._version, layout, A.device), blob)
        total += len(items)
    return total


class PackedOperands:
    """Per-module cache of the operand images, refreshed when A or B changed (torch version counters: the
    optimizer step, ``load_state_dict`` and ``copy_`` all bump them; re-pointing ``.data`` changes the address).
    The blob a step's forward used stays valid for its recompute and backward: a refresh allocates a new tensor
    whenever the previous one may still be referenced by a saved autograd context."""

    def __init__(self):
        self._held = {}          # DT_* code -> (stamp, blob)

    def get(self, A: torch.Tensor, B: torch.Tensor, layout: int, dtype=torch.bfloat16) -> torch.Tensor:
        dt = _dt_of(dtype)
        stamp = (A.data_ptr(), A._version, B.data_ptr(), B._version, layout, A.device)
        held = self._held.get(dt)
        if held is None or held[0] != stamp:
            held = (stamp, pack_operands(_master(A), _master(B), layout, dtype=dt))
            self._held[dt] = held
        return held[1]

    def invalidate(self) -> None:
        """Forget the images (after an in-place edit through ``.data`` that the version counters cannot see)."""
        self._held.clear()


class TransposedCopy:
    """``W^T`` (contiguous) of a FROZEN weight, kept beside it so that the input-gradient GEMM runs as
    ``F.linear(gy, W^T)`` (hipBLASLt's TN form) instead of ``gy @ W`` (NN): measured on MI355X at M = 41,472
    (tools/nn_vs_tn_probe.py) 349 vs 446 us for fc2, 296 vs 338 us for fc1, 77 vs 90 us for proj, 190 vs 225 us for qkv.
    Costs one extra copy of the frozen weights in HBM (0.9 GB for the SAM3 trunk in bf16)."""

    def __init__(self):
        self.t = None
        self._stamp = None

    def get(self, w: torch.Tensor) -> torch.Tensor:
        stamp = (w.data_ptr(), w._version, w.dtype, w.device, tuple(w.shape))
        if stamp != self._stamp:
            self.t = w.detach().t().contiguous()
            self._stamp = stamp
        return self.t


def _frozen_fwd(x2: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], x_q8=None) -> torch.Tensor:
    """The frozen layer's GEMM ``x2 @ w^T + b``: hipBLASLt bf16 / fp32, or the fp8 route of sam3_lora_amd.fp8 when enabled.
    ``x_q8`` = (e4m3 image of x2, its scale, id of the weight it was made for): x2's producer already quantised it."""
    from . import fp8
    if fp8.eligible(x2, w):
        if x_q8 is not None and x_q8[2] == id(w) and x_q8[0].shape == x2.shape:
            return fp8.fp8_linear_q(x_q8[0], x_q8[1], w, b)
        return fp8.fp8_linear(x2, w, b)
    return F.linear(x2, w, b)


def _fp8_companion(x: torch.Tensor):
    """The fp8 image a producing kernel attached to its bf16 output (vit.layer_norm_skip), if any."""
    return getattr(x, "_sam3_fp8", None)


def _hl_q8_ok(lora) -> bool:
    """The activation-fused adapter passes can carry an fp8 image: hi + lo kernels (rank <= 16, not single-rounded).  The LIBRARY
    is asked (its knob table is latched at first launch; re-reading the environment here could disagree with it, and the q8 entry
    points would then return ENOTSUP after the delayed-scaling state had already been flipped)."""
    rank = int(getattr(lora, "rank", 99))       # (the recompute form itself reaches r <= 32 since round 6; the fp8 image stays with r <= 16)
    return rank <= 16 and bool(_ffi.load().sam3_lora_bwd_act_recomputes_input(rank, DT_BF16, 0.0))


def _dx(gy2: torch.Tensor, w: torch.Tensor, wt: Optional[torch.Tensor]) -> torch.Tensor:
    """Input gradient of a frozen linear: ``gy2 @ w``, through the transposed copy when one of gy2's dtype is at hand."""
    from . import fp8
    if fp8.eligible(gy2, w):
        return fp8.fp8_dx(gy2, w)
    if wt is not None and wt.dtype == gy2.dtype and wt.shape == (w.shape[1], w.shape[0]):
        return F.linear(gy2, wt)
    return gy2 @ w


class _FrozenLinearFn(torch.autograd.Function):
    """``F.linear`` with a frozen weight whose backward uses the transposed copy (no weight / bias gradients)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wt, x_q8=None):
        ctx.save_for_backward(weight, wt)
        x2 = x.reshape(-1, x.shape[-1])
        return _frozen_fwd(x2 if x2.is_contiguous() else x2.contiguous(), weight, bias, x_q8).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        weight, wt = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.shape[-1])
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        return _dx(gy2, weight, wt).view(*gy.shape[:-1], weight.shape[1]), None, None, None, None


def frozen_linear(x: torch.Tensor, lin: torch.nn.Linear, cache: TransposedCopy) -> torch.Tensor:
    """``lin(x)``; a CUDA Linear with frozen parameters (and no autocast) gets the TN-form backward."""
    w = lin.weight
    from . import fp8
    frozen = x.is_cuda and not w.requires_grad and (lin.bias is None or not lin.bias.requires_grad) and x.dtype == w.dtype \
        and not torch.is_autocast_enabled("cuda")
    if frozen and x.requires_grad and torch.is_grad_enabled():
        # the transposed copy serves the bf16 / fp32 TN-form backward; the fp8 route keeps its own e4m3 transpose, so skip it
        # only for the layers that actually take that route
        return _FrozenLinearFn.apply(x, w, lin.bias, None if fp8.would_use(x, w) else cache.get(w), _fp8_companion(x))
    if frozen and fp8.fp8_enabled():        # no gradient needed (first block, eval): still the fp8 GEMM
        x2 = x.reshape(-1, x.shape[-1])
        return _frozen_fwd(x2 if x2.is_contiguous() else x2.contiguous(), w, lin.bias, _fp8_companion(x)).view(*x.shape[:-1], w.shape[0])
    return lin(x)


def lora_fwd_(x2: torch.Tensor, A: torch.Tensor, B: torch.Tensor, y2: torch.Tensor, scaling: float, layout: int,
              save_t: bool = False, drop_p: float = 0.0, seed: int = 0, offset: int = 0,
              packed: Optional[torch.Tensor] = None, gelu_out: Optional[torch.Tensor] = None, q8=None) -> Optional[torch.Tensor]:
    """In place: y2[M,out] += scaling * (x2[M,in] @ A_c) @ B_c.  Returns the saved-t blob if asked.
    ``gelu_out`` ([M,out], same dtype): additionally receives GELU(y2) from the same pass (sam3_lora_fwd_act).
    ``q8`` = ``fp8.producer_slots(...)``: GELU(y2) also leaves as an fp8 image (sam3_lora_fwd_act_q8; fp8 frozen-W mode)."""
    lib = _ffi.load()
    _require_cuda(x2, A, B, y2)
    M, fin = x2.shape
    fout = y2.shape[1]
    rank = _rank_of(A, layout)
    dt = _dtype_code(x2)
    if y2.dtype != x2.dtype:
        raise LoRAKernelError("sam3_lora_amd: x and y must share a dtype")
    nws = lib.sam3_lora_fwd_workspace_bytes(M, fin, fout, rank, dt)
    if nws == 0:
        raise LoRAKernelError(f"sam3_lora_fwd_workspace_bytes: {_ffi.last_error()}")
    ws = _workspace(x2.device, nws)
    tT = saved_t_like(M, rank, x2.device, dt) if save_t else None
    args = (x2.data_ptr(), (packed if packed is not None else A).data_ptr(), B.data_ptr(), y2.data_ptr(),
            tT.data_ptr() if tT is not None else None,
            M, fin, fout, rank, x2.stride(0), y2.stride(0), layout | (PREPACKED if packed is not None else 0), float(scaling),
            float(drop_p), int(seed), int(offset), dt, ws.data_ptr(), ws.numel(),
            ctypes.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream))
    if gelu_out is None:
        if q8 is not None:
            raise LoRAKernelError("sam3_lora_amd: an fp8 image rides on the GELU-fused pass only")
        _ffi.check(lib.sam3_lora_fwd(*args), "sam3_lora_fwd")
    else:
        if gelu_out.dtype != y2.dtype or gelu_out.shape != y2.shape or gelu_out.stride(1) != 1:
            raise LoRAKernelError("sam3_lora_amd: gelu_out must match y in shape and dtype")
        if q8 is None:
            _ffi.check(lib.sam3_lora_fwd_act(*args, _ffi.ACT_GELU, gelu_out.data_ptr(), gelu_out.stride(0)), "sam3_lora_fwd_act")
        else:
            img, fmt, a_in, a_out, sc = q8
            _ffi.check(lib.sam3_lora_fwd_act_q8(*args, _ffi.ACT_GELU, gelu_out.data_ptr(), gelu_out.stride(0), img.data_ptr(),
                                                img.stride(0), int(fmt), a_in.data_ptr(), a_out.data_ptr(), sc.data_ptr()),
                       "sam3_lora_fwd_act_q8")
    return tT


_FUSED = {"on": None}     # None: follow SAM3_LORA_FUSED_LINEAR in the environment (read once)


def set_fused_linear(on: Optional[bool]) -> None:
    """Route the fc1 -> GELU site of the fused MLP node through ``sam3_lora_linear_fwd`` (the adapter inside the frozen GEMM,
    SURVEY 8f-1) instead of hipBLASLt + ``sam3_lora_fwd_act``.  On by default where the kernel applies (bf16, rank <= 32, in_features a
    multiple of 64; measured on MI355X at the benchmark's fc1 site: 528 us against 584 us, profiles/r04a_fused_linear_probe.json;
    whole step 239.0 against 241.0 ms, profiles/r04c_bench_full_ab.json); ``SAM3_LORA_FUSED_LINEAR=0`` or ``set_fused_linear(False)``
    selects the two-pass form.  ``None`` restores the environment default."""
    _FUSED["on"] = on


def fused_linear_enabled() -> bool:
    if _FUSED["on"] is None:
        import os
        _FUSED["on"] = os.environ.get("SAM3_LORA_FUSED_LINEAR", "1") not in ("", "0")
    return bool(_FUSED["on"])


def set_fused_linear_fp8(on: Optional[bool]) -> None:
    """fp8 frozen-W mode: route the fc1 -> GELU site through ``sam3_lora_linear_fwd_q8`` (on, the default) or through hipBLASLt's
    fp8 GEMM + ``sam3_lora_fwd_act_q8`` (off); ``None`` restores the environment default (SAM3_LORA_FUSED_LINEAR_FP8)."""
    _FUSED["fp8"] = on


_KNOBS = {}


def _knob(name: str) -> bool:
    """A/B knobs of the fused MLP node, read from the environment once (``set_knob`` overrides)."""
    if name not in _KNOBS:
        import os
        _KNOBS[name] = os.environ.get(name, "0") not in ("", "0")
    return bool(_KNOBS[name])


def set_knob(name: str, on: Optional[bool]) -> None:
    if on is None:
        _KNOBS.pop(name, None)
    else:
        _KNOBS[name] = bool(on)


def _fused_layout_ok(x2: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
    """The layouts sam3_lora_linear_fwd addresses (16-byte aligned bases and row pitches, unit column stride of W, an 8-byte aligned
    bias, row pitches below 4 M elements); anything else keeps the two-pass form, which takes any strides."""
    e = x2.element_size()
    if W.dim() != 2 or W.stride(1) != 1 or x2.stride(1) != 1:
        return False
    if (x2.data_ptr() | W.data_ptr()) & 15 or (x2.stride(0) * e) & 15 or (W.stride(0) * e) & 15:
        return False
    if max(x2.stride(0), W.stride(0)) >= (1 << 22):
        return False
    return bias is None or (bias.is_contiguous() and bias.data_ptr() % 8 == 0)


def fused_linear_fp8_enabled() -> bool:
    """SAM3_LORA_FUSED_LINEAR_FP8=0: the fp8 frozen-W mode keeps hipBLASLt's fp8 GEMM + the adapter pass at the fc1 site."""
    if _FUSED.get("fp8") is None:
        import os
        _FUSED["fp8"] = os.environ.get("SAM3_LORA_FUSED_LINEAR_FP8", "1") not in ("", "0")
    return bool(_FUSED["fp8"])


def linear_fwd_supported(fin: int, fout: int, rank: int, dtype) -> bool:
    """Whether :func:`lora_linear_fwd_` takes this shape (sam3_lora_linear_fwd_supported)."""
    if dtype != torch.bfloat16:
        return False
    return bool(_ffi.load().sam3_lora_linear_fwd_supported(int(fin), int(fout), int(rank), DT_BF16))


def lora_linear_fwd_(x2: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], A: torch.Tensor, B: torch.Tensor,
                     scaling: float, layout: int, save_t: bool = False, drop_p: float = 0.0, seed: int = 0, offset: int = 0,
                     packed: Optional[torch.Tensor] = None, gelu: bool = False, y_out: Optional[torch.Tensor] = None,
                     gelu_out: Optional[torch.Tensor] = None):
    """The whole adapted Linear as ONE kernel (sam3_lora_linear_fwd): ``y = x2 @ W^T + bias + scaling * (drop(x2) @ A_c) @ B_c``
    with fp32 accumulation of all three terms and one rounding; ``gelu=True`` also returns GELU(y) from the same pass.
    Returns ``(y, gelu_out or None, saved-t blob or None)``."""
    lib = _ffi.load()
    _require_cuda(x2, W, A, B, bias)
    M, fin = x2.shape
    fout = W.shape[0]
    rank = _rank_of(A, layout)
    dt = _dtype_code(x2)
    if W.dtype != x2.dtype or (bias is not None and bias.dtype != x2.dtype) or W.shape[1] != fin or W.stride(1) != 1:
        raise LoRAKernelError("sam3_lora_amd: x, W and bias must share a dtype; W is [out, in] with unit column stride")
    if bias is not None and not bias.is_contiguous():
        bias = bias.contiguous()
    nws = lib.sam3_lora_linear_fwd_workspace_bytes(M, fin, fout, rank, dt)
    if nws == 0:
        raise LoRAKernelError(f"sam3_lora_linear_fwd_workspace_bytes: {_ffi.last_error() or 'shape / dtype not supported'}")
    ws = _workspace(x2.device, nws)
    y = y_out if y_out is not None else torch.empty(M, fout, dtype=x2.dtype, device=x2.device)
    a = None
    if gelu:
        a = gelu_out if gelu_out is not None else torch.empty_like(y)
    tT = saved_t_like(M, rank, x2.device, dt) if save_t else None
    rc = lib.sam3_lora_linear_fwd(
        x2.data_ptr(), W.data_ptr(), bias.data_ptr() if bias is not None else None,
        (packed if packed is not None else A).data_ptr(), B.data_ptr(), y.data_ptr(), tT.data_ptr() if tT is not None else None,
        M, fin, fout, rank, x2.stride(0), W.stride(0), y.stride(0), layout | (PREPACKED if packed is not None else 0),
        float(scaling), float(drop_p), int(seed), int(offset), dt, ws.data_ptr(), ws.numel(),
        ctypes.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream),
        _ffi.ACT_GELU if gelu else _ffi.ACT_NONE, a.data_ptr() if a is not None else None, a.stride(0) if a is not None else 0)
    _ffi.check(rc, "sam3_lora_linear_fwd")
    return y, a, tT


def lora_linear_fwd_q8_(x2: torch.Tensor, xq: torch.Tensor, sx: torch.Tensor, wq: torch.Tensor, sw: torch.Tensor,
                        bias: Optional[torch.Tensor], A: torch.Tensor, B: torch.Tensor, scaling: float, layout: int,
                        save_t: bool = False, drop_p: float = 0.0, seed: int = 0, offset: int = 0,
                        packed: Optional[torch.Tensor] = None, gelu: bool = False, q8=None):
    """:func:`lora_linear_fwd_` in the fp8 frozen-W mode (sam3_lora_linear_fwd_q8): the frozen GEMM on the e4m3 images ``xq`` [M, in]
    / ``wq`` [out, in] with their dequantisation scales ``sx`` / ``sw`` (one-element device tensors), the LoRA branch on the bf16
    ``x2``; ``q8`` = ``fp8.producer_slots(...)``: GELU(y) also leaves as the fp8 image of the next frozen GEMM.
    Returns ``(y, gelu_out or None, saved-t blob or None)``."""
    lib = _ffi.load()
    _require_cuda(x2, xq, wq, A, B, bias)
    M, fin = x2.shape
    fout = wq.shape[0]
    rank = _rank_of(A, layout)
    if x2.dtype != torch.bfloat16 or xq.shape != x2.shape or wq.shape[1] != fin or xq.stride(1) != 1 or wq.stride(1) != 1:
        raise LoRAKernelError("sam3_lora_amd: fused fp8 linear takes bf16 x [M, in], e4m3 images xq [M, in] and wq [out, in] with unit column stride")
    if xq.element_size() != 1 or wq.element_size() != 1:
        raise LoRAKernelError("sam3_lora_amd: xq / wq must be fp8 tensors")
    if bias is not None and (bias.dtype != torch.bfloat16 or not bias.is_contiguous()):
        bias = bias.to(torch.bfloat16).contiguous()
    nws = lib.sam3_lora_linear_fwd_workspace_bytes(M, fin, fout, rank, DT_BF16)
    if nws == 0:
        raise LoRAKernelError(f"sam3_lora_linear_fwd_workspace_bytes: {_ffi.last_error() or 'shape / dtype not supported'}")
    ws = _workspace(x2.device, nws)
    y = torch.empty(M, fout, dtype=x2.dtype, device=x2.device)
    a = torch.empty_like(y) if gelu else None
    tT = saved_t_like(M, rank, x2.device, DT_BF16) if save_t else None
    if q8 is not None and not gelu:
        raise LoRAKernelError("sam3_lora_amd: the fp8 image is the one of the activation output (gelu=True)")
    qimg, qfmt, am_in, am_out, qscale = q8 if q8 is not None else (None, 0, None, None, None)
    ptr = lambda t: t.data_ptr() if t is not None else None
    rc = lib.sam3_lora_linear_fwd_q8(
        x2.data_ptr(), xq.data_ptr(), xq.stride(0), sx.data_ptr(), wq.data_ptr(), wq.stride(0), sw.data_ptr(), ptr(bias),
        (packed if packed is not None else A).data_ptr(), B.data_ptr(), y.data_ptr(), ptr(tT),
        M, fin, fout, rank, x2.stride(0), y.stride(0), layout | (PREPACKED if packed is not None else 0),
        float(scaling), float(drop_p), int(seed), int(offset), DT_BF16, ws.data_ptr(), ws.numel(),
        ctypes.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream),
        _ffi.ACT_GELU if gelu else _ffi.ACT_NONE, ptr(a), a.stride(0) if a is not None else 0,
        ptr(qimg), qimg.stride(0) if qimg is not None else 0, int(qfmt), ptr(am_in), ptr(am_out), ptr(qscale))
    _ffi.check(rc, "sam3_lora_linear_fwd_q8")
    return y, a, tT


def lora_linear_dgrad_act_(gy2: torch.Tensor, Wt: torch.Tensor, A: torch.Tensor, B: torch.Tensor, scaling: float, layout: int,
                           gelu_pre: torch.Tensor) -> torch.Tensor:
    """Input gradient of an adapted Linear behind a GELU as ONE kernel (sam3_lora_linear_dgrad_act):
    ``(gy2 @ Wt^T + scaling * (gy2 @ B_c^T) @ A_c^T) * GELU'(gelu_pre)`` with ``Wt`` = the transposed frozen weight ``[in, out]``."""
    lib = _ffi.load()
    _require_cuda(gy2, Wt, A, B, gelu_pre)
    M, fout = gy2.shape
    fin = Wt.shape[0]
    rank = _rank_of(A, layout)
    if gy2.dtype != torch.bfloat16 or Wt.dtype != gy2.dtype or gelu_pre.dtype != gy2.dtype or Wt.shape[1] != fout or Wt.stride(1) != 1 \
            or gelu_pre.shape != (M, fin):
        raise LoRAKernelError("sam3_lora_amd: dgrad takes bf16 gy [M, out], Wt [in, out] with unit column stride and pre_act [M, in]")
    nws = lib.sam3_lora_linear_fwd_workspace_bytes(M, fout, fin, rank, DT_BF16)
    if nws == 0:
        raise LoRAKernelError(f"sam3_lora_linear_fwd_workspace_bytes: {_ffi.last_error() or 'shape / dtype not supported'}")
    ws = _workspace(gy2.device, nws)
    gx = torch.empty(M, fin, dtype=gy2.dtype, device=gy2.device)
    rc = lib.sam3_lora_linear_dgrad_act(
        gy2.data_ptr(), Wt.data_ptr(), A.data_ptr(), B.data_ptr(), gx.data_ptr(), M, fin, fout, rank, gy2.stride(0), Wt.stride(0),
        gx.stride(0), int(layout), float(scaling), DT_BF16, ws.data_ptr(), ws.numel(),
        ctypes.c_void_p(torch.cuda.current_stream(gy2.device).cuda_stream), _ffi.ACT_GELU, gelu_pre.data_ptr(), gelu_pre.stride(0))
    _ffi.check(rc, "sam3_lora_linear_dgrad_act")
    return gx


def lora_bwd_(gy2: torch.Tensor, x2: torch.Tensor, tT: Optional[torch.Tensor], A: torch.Tensor, B: torch.Tensor,
              gx2: Optional[torch.Tensor], gA: Optional[torch.Tensor], gB: Optional[torch.Tensor], scaling: float,
              layout: int, accumulate: bool = False, drop_p: float = 0.0, seed: int = 0, offset: int = 0,
              packed: Optional[torch.Tensor] = None, gelu_pre: Optional[torch.Tensor] = None, q8=None) -> None:
    """In place: gx2 += lora input-grad; gA/gB (fp32, caller layout) = or += the LoRA weight grads.
    ``gelu_pre`` ([M,in]): the pre-activation whose GELU produced x; gx2 leaves multiplied by GELU'(gelu_pre)
    (sam3_lora_bwd_act).  With ``gelu_pre`` given, ``x2`` may be None: "the input is GELU(gelu_pre)", recomputed inside
    that pass for gA (:func:`bwd_act_recomputes_input` says when the kernels can)."""
    lib = _ffi.load()
    if x2 is None:
        if gelu_pre is None or tT is None:
            raise LoRAKernelError("sam3_lora_amd: x may be None only with gelu_pre and the saved t^T given")
        shape_src = gelu_pre
    else:
        shape_src = x2
    _require_cuda(gy2, shape_src, A, B, gx2, gA, gB)
    M, fin = shape_src.shape
    fout = gy2.shape[1]
    rank = _rank_of(A, layout)
    dt = _dtype_code(shape_src)
    if gy2.dtype != shape_src.dtype or (gx2 is not None and gx2.dtype != shape_src.dtype):
        raise LoRAKernelError("sam3_lora_amd: gy, x and gx must share a dtype")
    for g in (gA, gB):
        if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
            raise LoRAKernelError("sam3_lora_amd: gA/gB must be contiguous float32")
    nws = lib.sam3_lora_bwd_workspace_bytes(M, fin, fout, rank, dt)
    if nws == 0:
        raise LoRAKernelError(f"sam3_lora_bwd_workspace_bytes: {_ffi.last_error()}")
    ws = _workspace(shape_src.device, nws)
    args = (gy2.data_ptr(), x2.data_ptr() if x2 is not None else None, tT.data_ptr() if tT is not None else None,
            (packed if packed is not None else A).data_ptr(), B.data_ptr(),
            gx2.data_ptr() if gx2 is not None else None,
            gA.data_ptr() if gA is not None else None, gB.data_ptr() if gB is not None else None,
            M, fin, fout, rank, gy2.stride(0), x2.stride(0) if x2 is not None else fin, gx2.stride(0) if gx2 is not None else fin,
            layout | (PREPACKED if packed is not None else 0), float(scaling), float(drop_p), int(seed), int(offset), dt,
            1 if accumulate else 0,
            ws.data_ptr(), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(shape_src.device).cuda_stream))
    if gelu_pre is None:
        if q8 is not None:
            raise LoRAKernelError("sam3_lora_amd: an fp8 image rides on the GELU'-fused pass only")
        _ffi.check(lib.sam3_lora_bwd(*args), "sam3_lora_bwd")
    else:
        if gx2 is None or gelu_pre.dtype != shape_src.dtype or gelu_pre.shape != shape_src.shape or gelu_pre.stride(1) != 1:
            raise LoRAKernelError("sam3_lora_amd: gelu_pre must match x in shape and dtype, and gx is required")
        if q8 is None:
            _ffi.check(lib.sam3_lora_bwd_act(*args, _ffi.ACT_GELU, gelu_pre.data_ptr(), gelu_pre.stride(0)), "sam3_lora_bwd_act")
        else:       # gx2 (the pre-activation gradient) also leaves as an fp8 image for the next input-gradient GEMM
            img, fmt, a_in, a_out, sc = q8
            _ffi.check(lib.sam3_lora_bwd_act_q8(*args, _ffi.ACT_GELU, gelu_pre.data_ptr(), gelu_pre.stride(0), img.data_ptr(),
                                                img.stride(0), int(fmt), a_in.data_ptr(), a_out.data_ptr(), sc.data_ptr()),
                       "sam3_lora_bwd_act_q8")


def bwd_act_recomputes_input(rank: int, dtype: torch.dtype, drop_p: float) -> bool:
    """Whether :func:`lora_bwd_` with ``gelu_pre`` accepts ``x2=None`` (sam3_lora_bwd_act_recomputes_input): the GELU' pass
    then recomputes GELU(gelu_pre) for gA, and the caller need not keep the activation for the backward."""
    if dtype not in (torch.bfloat16, torch.float32):
        return False
    return bool(_ffi.load().sam3_lora_bwd_act_recomputes_input(int(rank), 0 if dtype == torch.bfloat16 else 1, float(drop_p)))


def merge_weight(W: torch.Tensor, A: torch.Tensor, B: torch.Tensor, scaling: float, layout: int) -> torch.Tensor:
    """W + scaling * (A_c @ B_c)^T in fp32 on the GPU (lora_layer.py:81-88)."""
    lib = _ffi.load()
    _require_cuda(W, A, B)
    Wf, Af, Bf = W.detach().float().contiguous(), _master(A), _master(B)
    out = torch.empty_like(Wf)
    rc = lib.sam3_lora_merge(Wf.data_ptr(), Af.data_ptr(), Bf.data_ptr(), out.data_ptr(), Wf.shape[1], Wf.shape[0],
                             _rank_of(Af, layout), layout, float(scaling),
                             ctypes.c_void_p(torch.cuda.current_stream(W.device).cuda_stream))
    _ffi.check(rc, "sam3_lora_merge")
    return out


def _compute_dtype(x: torch.Tensor, weight: Optional[torch.Tensor]) -> torch.dtype:
    """The dtype the adapted layer computes in: the frozen weight's (bf16 layout or the reference's fp32), or the
    autocast dtype while autocast is on.  weight None: a bare LoRA branch (zero base) computes in x's dtype."""
    ref = weight if weight is not None else x
    cdt = ref.dtype if ref.dtype in (torch.bfloat16, torch.float32) else torch.float32
    if torch.is_autocast_enabled("cuda"):
        cdt = torch.get_autocast_dtype("cuda")
        if cdt not in (torch.bfloat16, torch.float32):
            raise LoRAKernelError(f"sam3_lora_amd: autocast dtype {cdt} unsupported (use bfloat16)")
    return cdt


# Direct accumulation of the weight gradients (opt-in, see enable_direct_grad_accumulation): `sam3_lora_bwd` adds
# straight into ``param.grad`` (accumulate = 1) instead of writing fresh gA / gB tensors for autograd to add.
_DIRECT = {"on": False}
_ZEROS = {}


def enable_direct_grad_accumulation(on: bool = True) -> None:
    """With this on, the backward of an adapted Linear whose ``lora_A.grad`` / ``lora_B.grad`` already exist (fp32,
    contiguous -- e.g. the views of :class:`sam3_lora_amd.ddp.LoRAGradReducer`'s flat buffer, or grads zeroed with
    ``zero_grad(set_to_none=False)``) accumulates into them inside the kernel's fixed-order reduction (accumulate = 1
    of the C-ABI): no gradient tensors are allocated and no separate overwrite-then-add happens.  Autograd still
    receives a gradient for A and B -- a stride-0 view of one shared zero scalar -- so its own bookkeeping stays exact:
    a parameter used several times in a graph is final only after its last use, post-accumulate hooks (the reducer's
    bucket launches) fire exactly once, tensor hooks see a (zero) gradient."""
    _DIRECT["on"] = bool(on)


class direct_grad_accumulation:
    """Scoped form of :func:`enable_direct_grad_accumulation`: ``with direct_grad_accumulation(): loss.backward()``.
    The switch changes what autograd is handed for A / B (a stride-0 zero while ``param.grad`` itself is updated inside
    the kernel), which is only right for ``backward()`` followed by an optimizer step -- so the trainer turns it on
    around exactly that call and restores the previous state afterwards; ``torch.autograd.grad``, tensor hooks and
    gradient-clipping hooks elsewhere in the process keep seeing real gradients.  ``touched`` collects the ids of the
    parameters whose ``.grad`` the kernels wrote inside the scope."""

    def __init__(self, on: bool = True):
        self.on, self.touched = bool(on), set()

    def __enter__(self):
        self._prev, self._prev_touched = _DIRECT["on"], _DIRECT.get("touched")
        _DIRECT["on"], _DIRECT["touched"] = self.on, self.touched
        return self

    def __exit__(self, *exc):
        _DIRECT["on"], _DIRECT["touched"] = self._prev, self._prev_touched
        return False


def _note_direct(*params) -> None:
    t = _DIRECT.get("touched")
    if t is not None:
        t.update(id(p) for p in params)


def _zero_grad_like(p: torch.Tensor) -> torch.Tensor:
    key = (p.device, p.dtype)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros((), device=p.device, dtype=p.dtype)
    return z.expand(p.shape)


def _is_master(p: torch.Tensor) -> bool:
    return p.dtype == torch.float32 and p.is_contiguous()


def _direct_targets(A: torch.Tensor, B: torch.Tensor):
    if not _DIRECT["on"]:
        return None
    for p in (A, B):
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or not g.is_cuda:
            return None
    return A.grad, B.grad


class _LoRALinearFn(torch.autograd.Function):
    """Frozen linear + LoRA branch as one autograd node (saved tensors: x, t^T -- never y or delta)."""

    @staticmethod
    def forward(ctx, x, weight, bias, A, B, scaling, layout, drop_p, seed, packed=None, wt=None):
        _require_cuda(x, weight, A, B)
        ctx.wt = wt
        cdt = _compute_dtype(x, weight)
        x2 = _rows(x if x.dtype == cdt else x.to(cdt))
        w = weight if (weight is None or weight.dtype == cdt) else weight.to(cdt)
        b = bias if (bias is None or bias.dtype == cdt) else bias.to(cdt)
        if w is None:
            fout = B.shape[1] if layout == LAYOUT_ROOT else B.shape[0]
            y2 = x2.new_zeros(x2.shape[0], fout)
        else:
            with torch.autocast("cuda", enabled=False):
                y2 = _frozen_fwd(x2, w, b)               # frozen GEMM: PyTorch-ROCm / hipBLASLt
        Am, Bm = _master(A), _master(B)
        need_w = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        fin, fout = x2.shape[1], y2.shape[1]
        ctx.pad = None
        ctx.packed = None
        tT = None
        if x2.shape[0] == 0:
            pass                                         # no rows (e.g. no geometric prompts): y is empty, grads are zero
        elif fin % 8 or fout % 8:
            # widths the 16-byte vector path cannot address (SAM3: geometry_encoder.*_direct_project, in = 2 / 4 /
            # 258): run the kernels on zero-padded copies.  The dropout counter then runs over the padded width.
            ctx.pad = (fin, fout, _pad8(fin), _pad8(fout))
            Ap, Bp = _pad_masters(Am, Bm, layout, *ctx.pad)
            x2 = _pad_cols(x2, ctx.pad[2])
            delta = x2.new_zeros(x2.shape[0], ctx.pad[3])
            tT = lora_fwd_(x2, Ap, Bp, delta, scaling, layout, save_t=need_w, drop_p=drop_p, seed=seed)
            y2 = y2 + delta[:, :fout]
        else:
            ctx.packed = packed
            tT = lora_fwd_(x2, Am, Bm, y2, scaling, layout, save_t=need_w, drop_p=drop_p, seed=seed, packed=packed)
        ctx.scaling, ctx.layout, ctx.drop_p, ctx.seed = scaling, layout, drop_p, seed
        ctx.x_shape, ctx.x_dtype = x.shape, x.dtype
        ctx.save_for_backward(x2, w, A, B, tT)
        return y2.view(*x.shape[:-1], y2.shape[-1])

    @staticmethod
    def backward(ctx, gy):
        x2, w, A, B, tT = ctx.saved_tensors
        gy2 = _rows(gy if gy.dtype == x2.dtype else gy.to(x2.dtype))
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        Am, Bm = _master(A), _master(B)
        gx2 = None
        if need_x:
            if w is None:
                gx2 = gy2.new_zeros(gy2.shape[0], ctx.x_shape[-1])
            else:
                with torch.autocast("cuda", enabled=False):
                    gx2 = _dx(gy2, w, ctx.wt)            # frozen GEMM (TN form when W^T is at hand)
        direct = _direct_targets(A, B) if (need_w and ctx.pad is None and gy2.shape[0] > 0 and _is_master(A) and _is_master(B)
                                           and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]) else None
        if direct is not None:
            _note_direct(A, B)
            lora_bwd_(gy2, x2, tT, Am, Bm, gx2, direct[0], direct[1], ctx.scaling, ctx.layout, accumulate=True,
                      drop_p=ctx.drop_p, seed=ctx.seed, packed=ctx.packed)
            gx = gx2.view(ctx.x_shape).to(ctx.x_dtype) if need_x else None
            return gx, None, None, _zero_grad_like(A), _zero_grad_like(B), None, None, None, None, None, None
        gA = torch.empty_like(Am) if need_w else None
        gB = torch.empty_like(Bm) if need_w else None
        if gy2.shape[0] == 0:
            if need_w:
                gA.zero_(), gB.zero_()
        elif ctx.pad is not None:
            fin, fout, fin_p, fout_p = ctx.pad
            Ap, Bp = _pad_masters(Am, Bm, ctx.layout, *ctx.pad)
            gxp = x2.new_zeros(x2.shape[0], fin_p) if need_x else None
            gAp = torch.empty_like(Ap) if need_w else None
            gBp = torch.empty_like(Bp) if need_w else None
            lora_bwd_(_pad_cols(gy2, fout_p), x2, tT, Ap, Bp, gxp, gAp, gBp, ctx.scaling, ctx.layout,
                      drop_p=ctx.drop_p, seed=ctx.seed)
            if need_x:
                gx2 = gx2 + gxp[:, :fin]
            if need_w:
                root = ctx.layout == LAYOUT_ROOT
                gA.copy_(gAp[:fin] if root else gAp[:, :fin])
                gB.copy_(gBp[:, :fout] if root else gBp[:fout])
        elif need_x or need_w:
            lora_bwd_(gy2, x2, tT, Am, Bm, gx2, gA, gB, ctx.scaling, ctx.layout,
                      drop_p=ctx.drop_p, seed=ctx.seed, packed=ctx.packed)
        gx = gx2.view(ctx.x_shape).to(ctx.x_dtype) if need_x else None
        if need_w:
            gA = gA.to(A.dtype) if ctx.needs_input_grad[3] else None
            gB = gB.to(B.dtype) if ctx.needs_input_grad[4] else None
        return gx, None, None, gA, gB, None, None, None, None, None, None


class _LoRAMlpFn(torch.autograd.Function):
    """``fc2(GELU(fc1(x)))`` with both Linears LoRA-adapted, as ONE autograd node: the GELU and its derivative ride
    on the adapters' in-place passes over the [M, hidden] tensor (``sam3_lora_fwd_act`` / ``sam3_lora_bwd_act``)
    instead of being elementwise kernels of their own.  Saved: x, the pre-activation h, the two t^T -- and a = GELU(h) only
    where the backward cannot recompute it inside its GELU' pass (dropout, rank > 16, fp32)."""

    @staticmethod
    def forward(ctx, x, W1, b1, A1, B1, s1, W2, b2, A2, B2, s2, layout, drop_p, seed1, seed2, pk1, pk2, Wt1=None, Wt2=None,
                x_q8=None, q8_ok=False):
        _require_cuda(x, W1, W2, A1, B1, A2, B2)
        from . import fp8
        cdt = W1.dtype
        x2 = _rows(x if x.dtype == cdt else x.to(cdt))
        need_w = any(ctx.needs_input_grad[i] for i in (3, 4, 8, 9))
        fused1 = (fused_linear_enabled() and not q8_ok and not fp8.eligible(x2, W1) and _fused_layout_ok(x2, W1, b1)
                  and linear_fwd_supported(W1.shape[1], W1.shape[0], _rank_of(_master(A1), layout), cdt))
        # (the layout predicate comes BEFORE anything with side effects: fp8.producer_slots below flips the delayed-scaling buffers of
        # W2's input role, so the fused call must be certain to be accepted once it has run)
        xq_given = x_q8 is not None and x_q8[2] == id(W1) and x_q8[0].shape == x2.shape
        fused1_q8 = (fused_linear_enabled() and fused_linear_fp8_enabled() and q8_ok and fp8.eligible(x2, W1) and x2.dtype == torch.bfloat16
                     and W1.shape[1] % 128 == 0 and _fused_layout_ok(x2, W1, b1)
                     and (not xq_given or (x_q8[0].stride(1) == 1 and x_q8[0].stride(0) % 16 == 0 and x_q8[0].data_ptr() % 16 == 0))
                     and linear_fwd_supported(W1.shape[1], W1.shape[0], _rank_of(_master(A1), layout), cdt))
        if fused1:
            # frozen GEMM + rank-r K step + bias + GELU in ONE kernel -- h and a are written once, never re-read
            qa = None
            h, a, t1 = lora_linear_fwd_(x2, W1, b1, _master(A1), _master(B1), s1, layout, save_t=need_w, drop_p=drop_p,
                                        seed=seed1, packed=pk1, gelu=True)
        elif fused1_q8:
            # the same in the fp8 frozen-W mode: e4m3 x (the LayerNorm's image when it made one for this weight) and W on the
            # scaled fp8 MFMA, the LoRA branch as a bf16 K step, and GELU(h) also as the e4m3 input of fc2's GEMM
            st = fp8.state_for(W1)
            if xq_given:
                xq, sx = x_q8[0], x_q8[1]
            else:
                xq, sx = st.qx(x2)
            qa = fp8.producer_slots(W2, "x", x2.shape[0], W1.shape[0], x2.device)
            h, a, t1 = lora_linear_fwd_q8_(x2, xq, sx, st.wq, st.scale, b1, _master(A1), _master(B1), s1, layout, save_t=need_w,
                                           drop_p=drop_p, seed=seed1, packed=pk1, gelu=True, q8=qa)
        else:
            with torch.autocast("cuda", enabled=False):
                h = _frozen_fwd(x2, W1, b1, x_q8)
            a = torch.empty_like(h)
            # fp8 frozen-W mode: GELU(h) leaves the adapter pass as bf16 (for fc2's adapter) AND as the e4m3 input of fc2's GEMM
            qa = fp8.producer_slots(W2, "x", h.shape[0], h.shape[1], h.device) if (q8_ok and h.dtype == torch.bfloat16) else None
            t1 = lora_fwd_(x2, _master(A1), _master(B1), h, s1, layout, save_t=need_w, drop_p=drop_p, seed=seed1, packed=pk1,
                           gelu_out=a, q8=qa)
        fused2 = (_knob("SAM3_LORA_FUSED_FC2") and fused_linear_enabled() and qa is None and not fp8.eligible(a, W2) and _fused_layout_ok(a, W2, b2)
                  and linear_fwd_supported(W2.shape[1], W2.shape[0], _rank_of(_master(A2), layout), cdt))
        ctx.q8_ok = q8_ok
        if fused2:
            # fc2 (K = 4736, N = 1024) through the same one-kernel form: an A/B knob -- 648 tiles = 2.5 rounds of a 256-CU grid and a
            # 74-step K loop lose to hipBLASLt's stream-K kernel here; off by default
            y, _, t2 = lora_linear_fwd_(a, W2, b2, _master(A2), _master(B2), s2, layout, save_t=need_w, drop_p=drop_p, seed=seed2,
                                        packed=pk2, gelu=False)
        else:
            with torch.autocast("cuda", enabled=False):
                y = fp8.fp8_linear_q(qa[0], qa[4], W2, b2) if qa is not None else _frozen_fwd(a, W2, b2)
            t2 = lora_fwd_(a, _master(A2), _master(B2), y, s2, layout, save_t=need_w, drop_p=drop_p, seed=seed2, packed=pk2)
        ctx.meta = (s1, s2, layout, drop_p, seed1, seed2, x.shape, x.dtype)
        ctx.pk = (pk1, pk2)
        ctx.wt = (Wt1, Wt2)
        # a = GELU(h) is fc2's input; the backward needs it for gA2 only, and where the kernels can they recompute it from h
        # inside the pass that applies GELU'(h) (sam3_lora_bwd_act with x == NULL): not saved then (393 MB per block at batch 8)
        # (the mirror only ever runs for an input gradient, on layouts the kernel addresses -- otherwise `a` would be kept for nothing)
        ctx.mirror = bool(_knob("SAM3_LORA_MIRROR") and ctx.needs_input_grad[0] and drop_p == 0.0 and h.dtype == torch.bfloat16 and not q8_ok
                          and not fp8.eligible(a, W2) and _fused_layout_ok(h, W2, None)
                          and (Wt2 is None or (Wt2.stride(1) == 1 and Wt2.data_ptr() % 16 == 0 and (Wt2.stride(0) * Wt2.element_size()) % 16 == 0))
                          and linear_fwd_supported(W2.shape[0], W2.shape[1], _rank_of(_master(A2), layout), cdt))
        ctx.recompute_a = bool(need_w and t2 is not None and not ctx.mirror
                               and bwd_act_recomputes_input(_rank_of(_master(A2), layout), h.dtype, drop_p))
        ctx.save_for_backward(x2, h, a.new_empty(0) if ctx.recompute_a else a, W1, W2, A1, B1, A2, B2, t1, t2)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, gy):
        x2, h, a, W1, W2, A1, B1, A2, B2, t1, t2 = ctx.saved_tensors
        s1, s2, layout, drop_p, seed1, seed2, x_shape, x_dtype = ctx.meta
        pk1, pk2 = ctx.pk
        need_x = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[i] for i in (3, 4, 8, 9))
        gy2 = _rows(gy if gy.dtype == x2.dtype else gy.to(x2.dtype))
        A1m, B1m, A2m, B2m = _master(A1), _master(B1), _master(A2), _master(B2)
        d1 = d2 = None
        if need_w and all(ctx.needs_input_grad[i] for i in (3, 4, 8, 9)) and all(_is_master(t) for t in (A1, B1, A2, B2)):
            d1, d2 = _direct_targets(A1, B1), _direct_targets(A2, B2)
        direct = d1 is not None and d2 is not None
        if direct:          # accumulate straight into param.grad (see enable_direct_grad_accumulation)
            (gA1, gB1), (gA2, gB2) = d1, d2
            _note_direct(A1, B1, A2, B2)
        else:
            gA1, gB1, gA2, gB2 = ((torch.empty_like(t) if need_w else None) for t in (A1m, B1m, A2m, B2m))
        from . import fp8
        if ctx.mirror and need_x and gy2.shape[0] > 0 and gy2.stride(1) == 1 and gy2.data_ptr() % 16 == 0 and (gy2.stride(0) * gy2.element_size()) % 16 == 0:
            # SURVEY 8f-1's backward mirror behind SAM3_LORA_MIRROR=1 (an A/B knob): gh as ONE MFMA kernel, the weight
            # gradients from the adapter backward without gx (it then reads the stored activation for gA)
            Wt2 = ctx.wt[1] if (ctx.wt[1] is not None and ctx.wt[1].dtype == gy2.dtype and ctx.wt[1].shape == (W2.shape[1], W2.shape[0])
                                and ctx.wt[1].stride(1) == 1) else W2.t().contiguous()
            ga = lora_linear_dgrad_act_(gy2, Wt2, A2m, B2m, s2, layout, gelu_pre=h)
            if need_w:
                lora_bwd_(gy2, a, t2, A2m, B2m, None, gA2, gB2, s2, layout, accumulate=direct, packed=pk2)
            gx2 = None
            with torch.autocast("cuda", enabled=False):
                gx2 = _dx(ga, W1, ctx.wt[0])
            lora_bwd_(ga, x2, t1, A1m, B1m, gx2, gA1, gB1, s1, layout, accumulate=direct, drop_p=drop_p, seed=seed1, packed=pk1)
            gx = gx2.view(x_shape).to(x_dtype)
            if direct:
                z = _zero_grad_like
                return (gx, None, None, z(A1), z(B1), None, None, None, z(A2), z(B2), None, None, None, None, None, None, None,
                        None, None, None, None)
            g = lambda t, p_, i: (t.to(p_.dtype) if (t is not None and ctx.needs_input_grad[i]) else None)
            return (gx, None, None, g(gA1, A1, 3), g(gB1, B1, 4), None, None, None, g(gA2, A2, 8), g(gB2, B2, 9), None, None,
                    None, None, None, None, None, None, None, None, None)
        with torch.autocast("cuda", enabled=False):
            ga = _dx(gy2, W2, ctx.wt[1])                             # frozen GEMM
        # fc2's adapter backward; its in-place pass over ga also applies GELU'(h): ga leaves as gh -- and, in the fp8 frozen-W
        # mode, as the e5m2 input of fc1's input-gradient GEMM
        qg = None
        if need_x and ctx.q8_ok and drop_p == 0.0 and ga.dtype == torch.bfloat16:
            qg = fp8.producer_slots(W1, "g", ga.shape[0], ga.shape[1], ga.device)
        lora_bwd_(gy2, None if ctx.recompute_a else a, t2, A2m, B2m, ga, gA2, gB2, s2, layout, accumulate=direct, drop_p=drop_p,
                  seed=seed2, packed=pk2, gelu_pre=h, q8=qg)
        gx2 = None
        if need_x:
            with torch.autocast("cuda", enabled=False):
                gx2 = fp8.fp8_dx_q(qg[0], qg[4], W1) if qg is not None else _dx(ga, W1, ctx.wt[0])     # frozen GEMM
        if need_x or need_w:
            lora_bwd_(ga, x2, t1, A1m, B1m, gx2, gA1, gB1, s1, layout, accumulate=direct, drop_p=drop_p, seed=seed1, packed=pk1)
        gx = gx2.view(x_shape).to(x_dtype) if need_x else None
        if direct:
            z = _zero_grad_like
            return (gx, None, None, z(A1), z(B1), None, None, None, z(A2), z(B2), None, None, None, None, None, None, None,
                    None, None, None, None)
        g = lambda t, p, i: (t.to(p.dtype) if (t is not None and ctx.needs_input_grad[i]) else None)
        return (gx, None, None, g(gA1, A1, 3), g(gB1, B1, 4), None, None, None, g(gA2, A2, 8), g(gB2, B2, 9), None, None,
                None, None, None, None, None, None, None, None, None)


def lora_mlp_gelu(x: torch.Tensor, fc1, fc2, layout: int, training: bool, wt_caches=None) -> Optional[torch.Tensor]:
    """``fc2(GELU(fc1(x)))`` for two LoRA-wrapped Linears through :class:`_LoRAMlpFn`, or None when the fused form does
    not apply (then the caller evaluates the three modules one by one).  ``fc1`` / ``fc2`` are
    ``(weight, bias, lora_layer)`` triples; the lora layers carry ``lora_A, lora_B, scaling, dropout_p, _packed``."""
    (W1, b1, l1), (W2, b2, l2) = fc1, fc2
    ok = (x.is_cuda and x.numel() > 0 and W1.dtype in (torch.bfloat16, torch.float32) and W2.dtype == W1.dtype
          and not W1.requires_grad and not W2.requires_grad and not torch.is_autocast_enabled("cuda")
          and x.dtype == W1.dtype and all(d % 8 == 0 for d in (*W1.shape, *W2.shape))
          and (b1 is None or b1.dtype == W1.dtype) and (b2 is None or b2.dtype == W1.dtype)
          and l1.dropout_p == l2.dropout_p)
    if not ok:
        return None
    p, seed1, seed2 = 0.0, 0, 0
    if training and l1.dropout_p > 0.0:
        p = float(l1.dropout_p)
        seed1, seed2 = (int(v) for v in torch.randint(0, 2 ** 62, (2,)).tolist())
    pk1 = l1._packed.get(l1.lora_A, l1.lora_B, int(layout), W1.dtype)
    pk2 = l2._packed.get(l2.lora_A, l2.lora_B, int(layout), W1.dtype)
    Wt1 = wt_caches[0].get(W1) if wt_caches is not None else None
    Wt2 = wt_caches[1].get(W2) if wt_caches is not None else None
    from . import fp8
    q8_ok = fp8.fp8_enabled() and _hl_q8_ok(l1) and _hl_q8_ok(l2)
    return _LoRAMlpFn.apply(x, W1, b1, l1.lora_A, l1.lora_B, float(l1.scaling), W2, b2, l2.lora_A, l2.lora_B,
                            float(l2.scaling), int(layout), p, seed1, seed2, pk1, pk2, Wt1, Wt2, _fp8_companion(x), q8_ok)


def lora_linear(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], A: torch.Tensor,
                B: torch.Tensor, scaling: float, layout: int, dropout_p: float = 0.0,
                training: bool = False, cache: Optional[PackedOperands] = None,
                wt_cache: Optional[TransposedCopy] = None) -> torch.Tensor:
    """``F.linear(x, weight, bias) + scaling * (dropout(x) @ A_c) @ B_c`` on the HIP path.

    ``layout`` selects how A/B are stored (LAYOUT_ROOT: A[in,r], B[r,out]; LAYOUT_PACKAGE:
    A[r,in], B[out,r]).  Dropout follows nn.Dropout semantics on the branch input only
    (lora_layers.py:54, lora_layer.py:73) and is generated inside the kernels from a counter-based
    hash; the seed is drawn from torch's CPU generator, so ``torch.manual_seed`` controls it and
    ``torch.utils.checkpoint`` (which restores the RNG state for its recompute) replays the mask.
    """
    p, seed = 0.0, 0
    if training and dropout_p > 0.0:
        p = float(dropout_p)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    packed = None
    if cache is not None and A.is_cuda and x.numel() > 0:
        fin = A.shape[0] if layout == LAYOUT_ROOT else A.shape[1]
        fout = B.shape[1] if layout == LAYOUT_ROOT else B.shape[0]
        if fin % 8 == 0 and fout % 8 == 0:
            packed = cache.get(A, B, int(layout), _compute_dtype(x, weight))
    wt = None
    if (wt_cache is not None and weight is not None and weight.is_cuda and not weight.requires_grad and x.requires_grad
            and torch.is_grad_enabled() and not torch.is_autocast_enabled("cuda") and weight.dtype == x.dtype):
        wt = wt_cache.get(weight)
    return _LoRALinearFn.apply(x, weight, bias, A, B, float(scaling), int(layout), p, seed, packed, wt)
