"""
ViT-Det trunk of the SAM3 image encoder -- the module that HOSTS the 64 adapted Linears
(SURVEY.md a11-a13).  Written from scratch for PyTorch-ROCm; same mathematics, module names,
parameter/buffer names and shapes as the reference (``sam3/model/vitdet.py``: PatchEmbed :299-336,
Attention :339-515, Block :518-613, ViT :616-859, window_partition :93-139, get_abs_pos :175-236,
2-D RoPE :40-90; timm ``Mlp``/``DropPath``), so a reference state dict loads with ``strict=True`` and
the reference injectors' module names (``...trunk.blocks.N.mlp.fc1``) are reproduced.

Differences in HOW (MI355X-first), none in WHAT:
  * qkv split + RoPE (real cos/sin tables, fp32) is one HIP pass; in window blocks the same pass gathers the tokens
    into windows and a fused residual kernel scatters them back, so no window (un)partition copy runs (the literal
    reshape+permute form remains for grids that need padding);
  * frozen LayerNorms run as one HIP pass each way; the MLP with both Linears adapted is one autograd node whose GELU
    and GELU' ride on the adapters' in-place passes (``functional.lora_mlp_gelu``);
  * frozen weights are meant to live in bf16, LoRA masters in fp32 (``to_training_layout``);
  * attention is ``F.scaled_dot_product_attention`` with the backend order measured fastest on MI355X;
  * per-block activation checkpointing is a policy, not a constant (``set_activation_checkpointing``): 288 GB of HBM
    hold the trunk's activations, and dropping the recompute removes a third of the step.

Only the configuration surface the SAM3 builder uses (``sam3/model_builder.py:69-96``) is supported:
no cls token retention, no relative-position bias, no LayerScale.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

__all__ = ["ViT", "Block", "Attention", "Mlp", "PatchEmbed", "DropPath", "sam3_vit", "to_training_layout",
           "set_activation_checkpointing", "layer_norm"]


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm semantics: keep mask ~ Bernoulli(1-p), scaled by 1/(1-p))."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (timm ``Mlp`` with the defaults the reference uses: no norm, drop=(p, 0))."""

    def __init__(self, in_features: int, hidden_features: int, drop: float = 0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop2 = nn.Dropout(0.0)

    def _adapted(self):
        """(layout, fc1 triple, fc2 triple) when both Linears carry a HIP LoRA adapter of the same family, else None."""
        from . import lora_layers as root_api
        from .functional import LAYOUT_PACKAGE, LAYOUT_ROOT
        from .lora import lora_layer as pkg_api
        f1, f2 = self.fc1, self.fc2
        if isinstance(f1, root_api.LoRALinear) and isinstance(f2, root_api.LoRALinear):
            return LAYOUT_ROOT, (f1.original_layer.weight, f1.original_layer.bias, f1.lora), \
                (f2.original_layer.weight, f2.original_layer.bias, f2.lora)
        if isinstance(f1, pkg_api.LinearWithLoRA) and isinstance(f2, pkg_api.LinearWithLoRA):
            return LAYOUT_PACKAGE, (f1.linear.weight, f1.linear.bias, f1.lora), (f2.linear.weight, f2.linear.bias, f2.lora)
        return None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # both Linears adapted, exact GELU, no activation dropout: one fused node -- the GELU and its derivative ride on
        # the adapters' in-place passes over [M, hidden] (functional.lora_mlp_gelu); otherwise module by module
        if (x.is_cuda and isinstance(self.act, nn.GELU) and self.act.approximate == "none"
                and isinstance(self.norm, nn.Identity) and not (self.training and (self.drop1.p > 0 or self.drop2.p > 0))):
            ad = self._adapted()
            if ad is not None:
                from .functional import lora_mlp_gelu
                y = lora_mlp_gelu(x, ad[1], ad[2], ad[0], self.training, wt_caches=(self.fc1._wt, self.fc2._wt))
                if y is not None:
                    return y
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class PatchEmbed(nn.Module):
    def __init__(self, patch_size: int, in_chans: int, embed_dim: int, bias: bool):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.proj(x).permute(0, 2, 3, 1)          # B C H W -> B H W C


def axial_rope_table(head_dim: int, size_x: int, size_y: int, theta: float, scale_pos: float) -> torch.Tensor:
    """complex64 [size_x*size_y, head_dim/2]: first half rotates with x, second half with y
    (reference ``compute_axial_cis``, vitdet.py:40-57)."""
    idx = torch.arange(0, head_dim, 4, device="cpu")[: head_dim // 4].float()
    freqs = 1.0 / (theta ** (idx / head_dim))
    t = torch.arange(size_x * size_y, dtype=torch.float32, device="cpu")
    tx = (t % size_x) * scale_pos
    ty = torch.div(t, size_x, rounding_mode="floor") * scale_pos
    ang = torch.cat([torch.outer(tx, freqs), torch.outer(ty, freqs)], dim=-1)
    return torch.polar(torch.ones_like(ang), ang)


class Attention(nn.Module):
    """Fused-qkv multi-head attention with 2-D axial RoPE over a (window or global) token grid."""

    def __init__(self, dim: int, num_heads: int, qkv_bias: bool, input_size: Tuple[int, int],
                 rope_theta: float = 10000.0, rope_pt_size: Optional[Tuple[int, int]] = None,
                 rope_interp: bool = False):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        pt = rope_pt_size if rope_pt_size is not None else input_size
        scale_pos = pt[0] / input_size[0] if rope_interp else 1.0
        # same buffer name/dtype as the reference so state dicts are interchangeable
        self.register_buffer("freqs_cis", axial_rope_table(self.head_dim, input_size[0], input_size[1],
                                                           rope_theta, scale_pos))
        self._cs = None   # (cos, sin) fp32 [L, head_dim/2], derived lazily on the right device
        from .functional import TransposedCopy
        self._wt_qkv, self._wt_proj = TransposedCopy(), TransposedCopy()   # W^T copies for the TN-form dX GEMMs

    def _cos_sin(self, device):
        if self._cs is None or self._cs[0].device != device:
            f = self.freqs_cis.to(device)
            self._cs = (f.real.float().contiguous(), f.imag.float().contiguous())
        return self._cs

    def _rope(self, t: torch.Tensor) -> torch.Tensor:
        """t [B, heads, L, head_dim]; adjacent pairs (2i, 2i+1) are one complex number."""
        cos, sin = self._cos_sin(t.device)
        tf = t.float().unflatten(-1, (-1, 2))
        a, b = tf[..., 0], tf[..., 1]
        out = torch.stack((a * cos - b * sin, a * sin + b * cos), dim=-1).flatten(-2)
        return out.to(t.dtype)

    def forward_windows(self, x: torch.Tensor, ws: int) -> torch.Tensor:
        """x [B, H, W, C] in IMAGE order -> attention inside ws x ws windows; returns the projected output as
        [B * nW, ws, ws, C] in WINDOW order.  qkv is a per-token Linear, so it runs on the image-order rows and the
        partition happens inside the qkv-split/RoPE kernel's addressing (no window_partition copy)."""
        B, H, W, C = x.shape
        L, nW = ws * ws, (H // ws) * (W // ws)
        qkv = self._lin(self.qkv, x, self._wt_qkv)
        cos, sin = self._cos_sin(qkv.device)
        q, k, v = _QKVRope.apply(qkv.reshape(B * H * W, 3 * C), cos, sin, B * nW, L, self.num_heads, self.head_dim,
                                 (ws, H, W))
        o = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        return self._lin(self.proj, o.transpose(1, 2).reshape(B * nW, ws, ws, C), self._wt_proj)

    @staticmethod
    def _lin(mod: nn.Module, x: torch.Tensor, cache) -> torch.Tensor:
        """``mod(x)``; a plain frozen nn.Linear gets the transposed-copy backward (functional.frozen_linear)."""
        if type(mod) is nn.Linear:
            from .functional import frozen_linear
            return frozen_linear(x, mod, cache)
        return mod(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, H, W, C = x.shape
        L = H * W
        qkv = self._lin(self.qkv, x, self._wt_qkv)
        if qkv.is_cuda and qkv.dtype in (torch.bfloat16, torch.float32):
            # one HIP pass: split q/k/v and rotate q, k; outputs are [B, L, heads, d] so SDPA gets
            # transposed VIEWS (no permute copies) and its output reshapes to [B, H, W, C] for free
            cos, sin = self._cos_sin(qkv.device)
            q, k, v = _QKVRope.apply(qkv.reshape(B * L, 3 * C), cos, sin, B, L, self.num_heads, self.head_dim)
            o = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
            o = o.transpose(1, 2).reshape(B, H, W, C)
        else:   # plain PyTorch formulation (CPU / other dtypes); same mathematics
            qkv = qkv.reshape(B, L, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
            q, k, v = qkv.unbind(0)
            q, k = self._rope(q), self._rope(k)
            o = F.scaled_dot_product_attention(q, k, v)
            o = o.permute(0, 2, 1, 3).reshape(B, H, W, C)
        return self._lin(self.proj, o, self._wt_proj)


_SDPA_ORDER = None


def _sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """``F.scaled_dot_product_attention`` with the backend order measured on MI355X / ROCm 7.2 / torch 2.10 for the
    trunk's two shapes (tools/sdpa_probe.py, fwd+bwd, bf16, head_dim 64): windows [72,16,576,64] 1.00 ms
    "efficient" vs 1.56 ms "flash"; global [8,16,5184,64] 5.4 ms vs 8.0 ms -- the default order picks flash."""
    global _SDPA_ORDER
    if not q.is_cuda:
        return F.scaled_dot_product_attention(q, k, v)
    from torch.nn.attention import SDPBackend, sdpa_kernel
    if _SDPA_ORDER is None:
        _SDPA_ORDER = [SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH]
    with sdpa_kernel(_SDPA_ORDER, set_priority=True):
        return F.scaled_dot_product_attention(q, k, v)


class _QKVRope(torch.autograd.Function):
    """qkv[B*L, 3*H*D] -> q, k (RoPE-rotated), v as [B, L, H, D]; C-ABI ``sam3_vit_qkv_rope_fwd/bwd``."""

    @staticmethod
    def forward(ctx, qkv2, cos, sin, B, L, H, D, win=(0, 0, 0)):
        """``win = (ws, Hh, Ww)``: qkv2 rows are image-order tokens, outputs are window-order (B windows of L)."""
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        qkv2 = qkv2.contiguous()
        q, k, v = (torch.empty(B, L, H, D, device=qkv2.device, dtype=qkv2.dtype) for _ in range(3))
        dt = 0 if qkv2.dtype == torch.bfloat16 else 1
        rc = lib.sam3_vit_qkv_rope_win_fwd(qkv2.data_ptr(), cos.data_ptr(), sin.data_ptr(), q.data_ptr(), k.data_ptr(),
                                           v.data_ptr(), B, L, H, D, win[0], win[1], win[2], dt,
                                           ctypes.c_void_p(torch.cuda.current_stream(qkv2.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sam3_vit_qkv_rope_win_fwd failed ({rc})")
        ctx.save_for_backward(cos, sin)
        ctx.dims = (B, L, H, D, dt, win)
        return q, k, v

    @staticmethod
    def backward(ctx, gq, gk, gv):
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        cos, sin = ctx.saved_tensors
        B, L, H, D, dt, win = ctx.dims
        # grads arrive [B, L, H, D]-shaped; view them as [B, H, L, D] with explicit strides for the kernel
        def canon(g):
            g = g if g.stride(-1) == 1 else g.contiguous()
            return g
        gq, gk, gv = canon(gq), canon(gk), canon(gv)
        if not (gq.stride() == gk.stride() == gv.stride()):
            gq, gk, gv = gq.contiguous(), gk.contiguous(), gv.contiguous()
        sb, sl, sh, _ = gq.stride()
        gqkv = torch.empty(B * L, 3 * H * D, device=gq.device, dtype=gq.dtype)
        rc = lib.sam3_vit_qkv_rope_win_bwd(gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), sb, sh, sl, cos.data_ptr(),
                                           sin.data_ptr(), gqkv.data_ptr(), B, L, H, D, win[0], win[1], win[2], dt,
                                           ctypes.c_void_p(torch.cuda.current_stream(gq.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sam3_vit_qkv_rope_win_bwd failed ({rc})")
        return gqkv, None, None, None, None, None, None, None


class _WinResidual(torch.autograd.Function):
    """``x + scale[b] * window_unpartition(h)`` in one pass (``scale``: per-image stochastic-depth factor or None);
    C-ABI ``sam3_vit_win_residual``.  The gradient of x is the incoming gradient itself (no kernel, no copy)."""

    @staticmethod
    def forward(ctx, x, h, scale, ws):
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        B, H, W, C = x.shape
        if h.dtype != x.dtype:          # the kernel reads both buffers with x's element type
            h = h.to(x.dtype)
        x, h = x.contiguous(), h.contiguous()
        y = torch.empty_like(x)
        dt = 0 if x.dtype == torch.bfloat16 else 1
        rc = lib.sam3_vit_win_residual(x.data_ptr(), h.data_ptr(), scale.data_ptr() if scale is not None else None,
                                       y.data_ptr(), B, H, W, C, ws, 0, dt,
                                       ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sam3_vit_win_residual failed ({rc})")
        if scale is not None:
            ctx.save_for_backward(scale)
        ctx.meta = (B, H, W, C, ws, dt, h.shape, scale is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        B, H, W, C, ws, dt, h_shape, has_scale = ctx.meta
        scale = ctx.saved_tensors[0] if has_scale else None
        gy = gy.contiguous()
        gh = None
        if ctx.needs_input_grad[1]:
            gh = torch.empty(h_shape, device=gy.device, dtype=gy.dtype)
            rc = lib.sam3_vit_win_residual(gy.data_ptr(), None, scale.data_ptr() if scale is not None else None,
                                           gh.data_ptr(), B, H, W, C, ws, 1, dt,
                                           ctypes.c_void_p(torch.cuda.current_stream(gy.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"sam3_vit_win_residual (backward) failed ({rc})")
        return (gy if ctx.needs_input_grad[0] else None), gh, None, None


class _FrozenLayerNorm(torch.autograd.Function):
    """LayerNorm with frozen weight / bias: one HIP pass each way (``sam3_vit_layernorm_fwd/bwd``), saving x and
    the fp32 row statistics; no parameter gradients are produced."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, q8=None):
        """``q8`` = ``fp8.producer_slots(...)`` of the frozen GEMM that consumes the output: its e4m3 image is written by the
        same pass (fp8 frozen-W mode, ``sam3_vit_layernorm_fwd_q8``)."""
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M = x2.shape[0]
        y = torch.empty_like(x2)
        stats = torch.empty(2, M, device=x.device, dtype=torch.float32)
        dt = 0 if x.dtype == torch.bfloat16 else 1
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if q8 is not None and dt == 0:
            img, fmt, a_in, a_out, sc = q8
            rc = lib.sam3_vit_layernorm_fwd_q8(x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), stats[0].data_ptr(),
                                               stats[1].data_ptr(), M, C, float(eps), dt, img.data_ptr(), img.stride(0), int(fmt),
                                               a_in.data_ptr(), a_out.data_ptr(), sc.data_ptr(), st)
        else:
            rc = lib.sam3_vit_layernorm_fwd(x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), stats[0].data_ptr(),
                                            stats[1].data_ptr(), M, C, float(eps), dt, st)
        if rc != 0:
            raise RuntimeError(f"sam3_vit_layernorm_fwd failed ({rc})")
        ctx.save_for_backward(x2, weight, stats)
        ctx.meta = (M, C, dt, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        x2, weight, stats = ctx.saved_tensors
        M, C, dt, shape = ctx.meta
        gy2 = gy.reshape(M, C)
        gy2 = gy2 if gy2.is_contiguous() else gy2.contiguous()
        gx = torch.empty_like(x2)
        rc = lib.sam3_vit_layernorm_bwd(gy2.data_ptr(), x2.data_ptr(), weight.data_ptr(), stats[0].data_ptr(),
                                        stats[1].data_ptr(), gx.data_ptr(), M, C, dt,
                                        ctypes.c_void_p(torch.cuda.current_stream(gy.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sam3_vit_layernorm_bwd failed ({rc})")
        return gx.view(shape), None, None, None, None


class _FrozenLayerNormSkip(torch.autograd.Function):
    """``x -> (x, LayerNorm(x))`` for a tensor that feeds both the norm and the residual around it: the backward adds
    the skip-path gradient inside the LayerNorm-backward pass (``sam3_vit_layernorm_bwd_add``) instead of leaving the
    sum of the two gradients to a separate accumulation kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, q8=None):
        y = _FrozenLayerNorm.forward(ctx, x, weight, bias, eps, q8)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, gskip, gy):
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        x2, weight, stats = ctx.saved_tensors
        M, C, dt, shape = ctx.meta
        if gy is None:
            return gskip, None, None, None, None
        gy2 = gy.reshape(M, C)
        gy2 = gy2 if gy2.is_contiguous() else gy2.contiguous()
        add = None
        if gskip is not None:
            add = gskip.reshape(M, C)
            add = add if add.is_contiguous() else add.contiguous()
        gx = torch.empty_like(x2)
        rc = lib.sam3_vit_layernorm_bwd_add(gy2.data_ptr(), x2.data_ptr(), weight.data_ptr(), stats[0].data_ptr(),
                                            stats[1].data_ptr(), add.data_ptr() if add is not None else None, gx.data_ptr(),
                                            M, C, dt, ctypes.c_void_p(torch.cuda.current_stream(gy.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sam3_vit_layernorm_bwd_add failed ({rc})")
        return gx.view(shape), None, None, None, None


def _ln_fast(norm: nn.Module, x: torch.Tensor) -> bool:
    return (isinstance(norm, nn.LayerNorm) and x.is_cuda and norm.elementwise_affine and norm.bias is not None
            and len(norm.normalized_shape) == 1 and not norm.weight.requires_grad and not norm.bias.requires_grad
            and x.dtype in (torch.bfloat16, torch.float32) and norm.weight.dtype == x.dtype and norm.bias.dtype == x.dtype
            and x.shape[-1] % 8 == 0 and x.shape[-1] <= 4096 and x.numel() > 0 and not torch.is_autocast_enabled("cuda"))


def _q8_for(consumer_weight, x: torch.Tensor):
    """fp8 frozen-W mode: the slots with which the LayerNorm kernel writes the e4m3 image of its output for the frozen GEMM
    with weight ``consumer_weight`` (None: mode off / not that kind of consumer / first use of the role)."""
    if consumer_weight is None or x.dtype != torch.bfloat16:
        return None
    from . import fp8
    if not fp8.fp8_enabled():
        return None
    C = x.shape[-1]
    return fp8.producer_slots(consumer_weight, "x", x.numel() // C, C, x.device)


def _attach_q8(y: torch.Tensor, q8, consumer_weight) -> torch.Tensor:
    if q8 is not None:       # read by functional.frozen_linear / lora_mlp_gelu when y reaches the consumer unchanged
        y._sam3_fp8 = (q8[0], q8[4], id(consumer_weight))
    return y


def layer_norm_skip(norm: nn.Module, x: torch.Tensor, consumer_weight=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(x, norm(x))`` -- use the returned x for the residual so that both gradients meet inside one kernel.
    ``consumer_weight``: the frozen weight of the Linear the normalised tensor feeds (fp8 frozen-W mode, see `_q8_for`)."""
    if _ln_fast(norm, x) and x.requires_grad and torch.is_grad_enabled():
        q8 = _q8_for(consumer_weight, x)
        xs, y = _FrozenLayerNormSkip.apply(x, norm.weight, norm.bias, norm.eps, q8)
        return xs, _attach_q8(y, q8, consumer_weight)
    return x, layer_norm(norm, x, consumer_weight)


def layer_norm(norm: nn.Module, x: torch.Tensor, consumer_weight=None) -> torch.Tensor:
    """``norm(x)``; an ``nn.LayerNorm`` over the last dimension whose parameters are frozen and share x's dtype runs
    on the HIP kernels, everything else (Identity, trainable or mixed-dtype norms, CPU) on the module itself."""
    if _ln_fast(norm, x):
        q8 = _q8_for(consumer_weight, x)
        return _attach_q8(_FrozenLayerNorm.apply(x, norm.weight, norm.bias, norm.eps, q8), q8, consumer_weight)
    return norm(x)


def window_partition(x: torch.Tensor, ws: int) -> Tuple[torch.Tensor, Tuple[int, int]]:
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win: torch.Tensor, ws: int, pad_hw: Tuple[int, int], hw: Tuple[int, int]) -> torch.Tensor:
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // ((Hp // ws) * (Wp // ws))
    x = win.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :] if (Hp > H or Wp > W) else x


class Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float, qkv_bias: bool, drop_path: float,
                 window_size: int, input_size: Tuple[int, int], rope_pt_size: Tuple[int, int],
                 rope_interp: bool, dropout: float = 0.0, eps: float = 1e-5):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias,
                              input_size=input_size if window_size == 0 else (window_size, window_size),
                              rope_pt_size=rope_pt_size, rope_interp=rope_interp)
        self.ls1 = nn.Identity()
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop=dropout)
        self.ls2 = nn.Identity()
        self.dropout = nn.Dropout(dropout)
        self.window_size = window_size

    def _fused_windows(self, x: torch.Tensor) -> bool:
        ws = self.window_size
        # not under autocast: norm1 would stay fp32 while qkv / SDPA / proj turn bf16, and the residual kernel takes
        # ONE dtype for both of its inputs
        return (ws > 0 and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.shape[1] % ws == 0
                and not torch.is_autocast_enabled("cuda")
                and x.shape[2] % ws == 0 and x.shape[-1] % 8 == 0
                and not (self.training and isinstance(self.dropout, nn.Dropout) and self.dropout.p > 0.0))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._fused_windows(x):
            # window blocks without partition / unpartition copies: rows stay in image order through norm1 and qkv,
            # the qkv-split/RoPE kernel gathers them into windows, and the residual add scatters them back
            xs, h = layer_norm_skip(self.norm1, x, self._qkv_weight())
            hw = self.attn.forward_windows(h, self.window_size)
            x = _WinResidual.apply(xs, hw, self._drop_path_scale(x), self.window_size)
            xs, h = layer_norm_skip(self.norm2, x, self._fc1_weight())
            return self._residual(xs, self.mlp(h))
        xs, h = layer_norm_skip(self.norm1, x, self._qkv_weight() if self.window_size == 0 else None)
        if self.window_size > 0:
            H, W = h.shape[1], h.shape[2]
            h, pad_hw = window_partition(h, self.window_size)
        h = self.attn(h)
        if self.window_size > 0:
            h = window_unpartition(h, self.window_size, pad_hw, (H, W))
        x = self._residual(xs, h)
        xs, h = layer_norm_skip(self.norm2, x, self._fc1_weight())
        return self._residual(xs, self.mlp(h))

    def _qkv_weight(self):
        """The frozen weight norm1's output feeds directly (fp8 frozen-W mode: LayerNorm then writes its e4m3 image)."""
        q = self.attn.qkv
        return q.weight if type(q) is nn.Linear and not q.weight.requires_grad else None

    def _fc1_weight(self):
        ad = self.mlp._adapted() if isinstance(self.mlp, Mlp) else None
        if ad is not None:
            return ad[1][0]
        f1 = getattr(self.mlp, "fc1", None)
        return f1.weight if type(f1) is nn.Linear and not f1.weight.requires_grad else None

    def _drop_path_scale(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """fp32 [B]: Bernoulli(keep) / keep per image while stochastic depth is active, else None."""
        dp = self.drop_path
        if isinstance(dp, DropPath) and dp.drop_prob > 0.0 and self.training:
            keep = 1.0 - dp.drop_prob
            mask = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
            if keep > 0.0 and dp.scale_by_keep:
                mask.div_(keep)
            return mask
        return None

    def _residual(self, x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
        """x + dropout(drop_path(h)); with stochastic depth active the mask-multiply and the add are one
        fused pass (addcmul) instead of two elementwise kernels over [B, 72, 72, C]."""
        h = self.dropout(h)
        dp = self.drop_path
        if isinstance(dp, DropPath) and dp.drop_prob > 0.0 and self.training:
            keep = 1.0 - dp.drop_prob
            mask = h.new_empty((h.shape[0],) + (1,) * (h.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and dp.scale_by_keep:
                mask.div_(keep)
            return torch.addcmul(x, h, mask)
        return x + h


class ViT(nn.Module):
    """SAM3 ViT-Det trunk.  ``forward(images[B,3,S,S]) -> [features[B, C, S/p, S/p]]``."""

    def __init__(self, img_size: int = 1008, pretrain_img_size: int = 336, patch_size: int = 14,
                 in_chans: int = 3, embed_dim: int = 1024, depth: int = 32, num_heads: int = 16,
                 mlp_ratio: float = 4.625, qkv_bias: bool = True, drop_path_rate: float = 0.1,
                 window_size: int = 24, global_att_blocks: Sequence[int] = (7, 15, 23, 31),
                 use_interp_rope: bool = True, pretrain_use_cls_token: bool = True, tile_abs_pos: bool = True,
                 ln_pre: bool = True, ln_post: bool = False, bias_patch_embed: bool = False,
                 use_act_checkpoint: bool = True, dropout: float = 0.0):
        super().__init__()
        self.pretrain_use_cls_token = pretrain_use_cls_token
        self.tile_abs_pos = tile_abs_pos
        self.full_attn_ids = list(global_att_blocks)
        self.use_act_checkpoint = use_act_checkpoint
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, bias_patch_embed)
        n_pos = (pretrain_img_size // patch_size) ** 2 + (1 if pretrain_use_cls_token else 0)
        self.pos_embed = nn.Parameter(torch.zeros(1, n_pos, embed_dim))
        grid = img_size // patch_size
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, dpr[i],
                  window_size=0 if i in self.full_attn_ids else window_size,
                  input_size=(grid, grid), rope_pt_size=(window_size, window_size),
                  rope_interp=use_interp_rope, dropout=dropout)
            for i in range(depth)])
        self.ln_pre = nn.LayerNorm(embed_dim, eps=1e-5) if ln_pre else nn.Identity()
        self.ln_post = nn.LayerNorm(embed_dim, eps=1e-5) if ln_post else nn.Identity()
        self.channel_list = [embed_dim]
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.apply(self._init_weights)
        self._pos_cache = {}

    @staticmethod
    def _init_weights(m: nn.Module) -> None:
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def abs_pos(self, h: int, w: int) -> torch.Tensor:
        """[1, h, w, C]: the pretrain grid (cls dropped) tiled -- or bicubically resized -- to (h, w)."""
        pe = self.pos_embed[:, 1:] if self.pretrain_use_cls_token else self.pos_embed
        size = int(math.sqrt(pe.shape[1]))
        if size == h and size == w:
            return pe.reshape(1, h, w, -1)
        g = pe.reshape(1, size, size, -1).permute(0, 3, 1, 2)
        if self.tile_abs_pos:
            g = g.tile([1, 1, h // size + 1, w // size + 1])[:, :, :h, :w]
        else:
            g = F.interpolate(g, size=(h, w), mode="bicubic", align_corners=False)
        return g.permute(0, 2, 3, 1)

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        x = self.patch_embed(x)
        h, w = x.shape[1], x.shape[2]
        self._last_tokens_per_image = h * w
        x = x + self.abs_pos(h, w).to(x.dtype)
        x = layer_norm(self.ln_pre, x)
        outs = []
        for i, blk in enumerate(self.blocks):
            if self.use_act_checkpoint and self.training:
                x = checkpoint(blk, x, use_reentrant=False)
            else:
                x = blk(x)
            if i == self.full_attn_ids[-1]:
                x = layer_norm(self.ln_post, x)
                outs.append(x.permute(0, 3, 1, 2))
        return outs


def sam3_vit(**overrides) -> ViT:
    """The one SAM3 image-encoder trunk (``sam3/model_builder.py:69-96``): 1008/14 = 72x72 tokens,
    1024-d, 32 blocks, 16 heads, MLP 4736, window 24, global blocks 7/15/23/31, interpolated RoPE."""
    return ViT(**overrides)


# Sub-modules of the SAM3 image model that stay in fp32 inside the bf16 training layout ("islands"): the DETR decoder with its
# box / presence heads and the scoring head -- 21 M parameters on 200-400 queries, a negligible share of the step's bytes and
# FLOPs, and the place where the matcher's cost and the box regression are formed.  torch.autocast, the reference's own
# mixed-precision mode, keeps exactly this kind of state (residuals, LayerNorm, small heads) in fp32.  Measured on MI355X against the
# reference's fp32 run: profiles/r04e_bf16_islands.json.
DEFAULT_FP32_ISLANDS = ("transformer.decoder", "dot_prod_scoring")
# ... except the memory-side work inside the decoder ("holes", fnmatch patterns): each layer's IMAGE cross-attention (K / V projections
# of all 5184 x batch memory tokens, the SDPA against them) and the two MLPs of the box-relative position bias that feeds it (a
# [batch x heads, queries, 5184] tensor per layer: 266 MB in bf16).  In fp32 these cost 4 % of the whole step (250.7 against 241 ms,
# profiles/r04f_bench_full_islands.json); the query stream they update -- 200-400 rows -- keeps its fp32 residuals, norms, self-
# attention, FFN and heads, so the rounding of a layer's attention output enters as a bf16-relative error of that UPDATE only.
DEFAULT_FP32_HOLES = ("transformer.decoder.layers.*.cross_attn", "transformer.decoder.boxRPB_embed_x", "transformer.decoder.boxRPB_embed_y")
_ISLAND_CONSUMERS = ("segmentation_head",)      # bf16 modules that take an island's fp32 outputs (the decoder's queries)
_ISLAND_ENTRY_SKIP = {"transformer.decoder": ("memory", "pos")}     # arguments only the holes consume: not cast at the island's entry


def _tree_cast(obj, dtype):
    if isinstance(obj, torch.Tensor):
        if obj.dtype in (torch.float32, torch.bfloat16) and obj.dtype != dtype:
            return obj.to(dtype)
        return obj
    if isinstance(obj, tuple) and hasattr(obj, "_fields"):        # namedtuple: positional constructor
        return type(obj)(*(_tree_cast(o, dtype) for o in obj))
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tree_cast(o, dtype) for o in obj)
    if isinstance(obj, dict):
        return {k: _tree_cast(v, dtype) for k, v in obj.items()}
    return obj


def _cast_inputs_hook(dtype, skip=()):
    """Forward pre-hook casting floating-point tensor arguments to ``dtype``; arguments named in ``skip`` are left alone whether they
    arrive by keyword or by position (bound through the module's forward signature: a positional ``memory`` would otherwise get an fp32
    copy of all 5184 x batch memory tokens at the decoder's entry)."""
    positional = {}          # type(module) -> positional parameter names of its forward, resolved once (not per call: this hook runs
                             # on every forward of every boundary module of a launch-bound step)

    def hook(module, args, kwargs):
        skip_pos = ()
        if skip and args:
            names = positional.get(type(module))
            if names is None:
                import inspect
                try:
                    names = tuple(inspect.signature(module.forward).parameters)
                except (TypeError, ValueError):
                    names = ()
                positional[type(module)] = names
            skip_pos = tuple(i for i, n in enumerate(names[:len(args)]) if n in skip)
        new_args = tuple(a if i in skip_pos else _tree_cast(a, dtype) for i, a in enumerate(args))
        return new_args, {k: (v if k in skip else _tree_cast(v, dtype)) for k, v in kwargs.items()}
    return hook


def to_training_layout(model: nn.Module, frozen_dtype: torch.dtype = torch.bfloat16, fp32_islands=None, fp32_holes=None) -> nn.Module:
    """MI355X training layout: every frozen tensor in bf16, trainable (LoRA) tensors stay fp32.  ``fp32_islands``: name prefixes of
    sub-modules whose frozen tensors stay fp32 (default :data:`DEFAULT_FP32_ISLANDS`; ``()`` = none), ``fp32_holes``: fnmatch patterns
    of sub-modules INSIDE an island that follow ``frozen_dtype`` after all (default :data:`DEFAULT_FP32_HOLES`).  Floating-point inputs
    are cast at every such boundary (forward pre-hooks), and so are the inputs of the bf16 modules that consume an island's outputs."""
    import fnmatch
    islands = tuple(DEFAULT_FP32_ISLANDS if fp32_islands is None else fp32_islands)
    mods = dict(model.named_modules())
    missing = tuple(i for i in islands if i not in mods)
    if missing and len(missing) == len(islands) and any(n == "transformer" or n.endswith(".transformer") for n in mods):
        # e.g. the model wrapped under a `module.` / `detector.` prefix: dropping the names silently would give the all-bf16 layout
        # (round 3's precision: presence logit 6e-2 instead of 8e-3) without anyone noticing
        import warnings
        warnings.warn(f"to_training_layout: fp32 islands {missing} are not sub-modules of this model (prefix the names as in "
                      f"named_modules()); those parts follow {frozen_dtype}", stacklevel=2)
    islands = tuple(i for i in islands if i in mods)
    hole_pats = tuple(DEFAULT_FP32_HOLES if fp32_holes is None else fp32_holes)
    holes = tuple(n for n in mods if any(fnmatch.fnmatchcase(n, p) for p in hole_pats)
                  and any(n.startswith(i + ".") for i in islands))

    def kept(name: str) -> bool:
        if any(name == h or name.startswith(h + ".") for h in holes):
            return False
        return any(name == i or name.startswith(i + ".") for i in islands)
    for name, p in model.named_parameters():
        if not p.requires_grad and p.dtype.is_floating_point and not kept(name):
            p.data = p.data.to(frozen_dtype)
    for name, b in model.named_buffers():
        if b.dtype.is_floating_point and not b.dtype.is_complex and not kept(name):
            b.data = b.data.to(frozen_dtype)
    for h in getattr(model, "_sam3_layout_hooks", ()):
        h.remove()
    hooks = []
    if islands and frozen_dtype != torch.float32:
        for i in islands:
            skip = _ISLAND_ENTRY_SKIP.get(i, ()) if any(h.startswith(i + ".") for h in holes) else ()
            hooks.append(mods[i].register_forward_pre_hook(_cast_inputs_hook(torch.float32, skip), with_kwargs=True))
        for h in holes:
            hooks.append(mods[h].register_forward_pre_hook(_cast_inputs_hook(frozen_dtype), with_kwargs=True))
        for c in _ISLAND_CONSUMERS:
            if c in mods and not kept(c) and mods[c] is not None:
                hooks.append(mods[c].register_forward_pre_hook(_cast_inputs_hook(frozen_dtype), with_kwargs=True))
    model._sam3_layout_hooks = hooks
    model._sam3_fp32_islands = islands
    model._sam3_fp32_holes = holes
    return model


def set_activation_checkpointing(model: nn.Module, mode="auto", batch: int = 8, headroom: float = 0.5) -> bool:
    """Per-block activation checkpointing of every :class:`ViT` inside ``model``: ``True`` / ``False`` / ``"auto"``.

    The reference recomputes each block in backward (``vitdet.py:837-838``) because its target GPUs hold 24-80 GB.
    A block of the SAM3 trunk keeps about 14 tensors of ``[tokens, C]`` alive for backward (measured: 1.17 GB per block
    at batch 8, bf16 -> 37 GB for the trunk); with 288 GB of HBM3E that fits many times over, and dropping the recompute
    removes a third of the step (217 -> 157 ms at batch 8 on MI355X).  ``"auto"`` keeps checkpointing only when the
    estimate for ``batch`` images exceeds ``headroom`` x the device's currently free memory.  Results are identical
    either way (the recompute replays the same RNG state).  Returns the setting applied.
    """
    vits = [m for m in model.modules() if isinstance(m, ViT)]
    if mode == "auto":
        use = True
        p = next((p for v in vits for p in v.parameters()), None)
        if p is not None and p.is_cuda:
            need = 0
            for v in vits:
                C = v.patch_embed.proj.out_channels
                grid = getattr(v, "_last_tokens_per_image", 72 * 72)     # tokens per image seen by the last forward
                need += len(v.blocks) * 16 * batch * grid * C * 2         # ~16 saved [tokens, C] tensors per block, bf16
            free, _ = torch.cuda.mem_get_info(p.device)
            use = need > headroom * free
    else:
        use = bool(mode)
    for v in vits:
        v.use_act_checkpoint = use
    # the layers around the trunk (this library's SAM3 restatement) follow the same setting: their activations are
    # a few GB at batch 8 (5184 image tokens x 256 channels x 12 layers)
    flags = {"TransformerEncoderFusion": "use_act_checkpoint", "TransformerDecoder": "use_act_checkpoint",
             "Transformer": "grad_checkpointing", "UniversalSegmentationHead": "act_ckpt",
             "Sam3Image": "use_act_checkpoint_seg_head", "SequenceGeometryEncoder": "use_act_ckpt"}
    for m in model.modules():
        attr = flags.get(type(m).__name__)
        if attr is not None and type(m).__module__.startswith("sam3_lora_amd.") and hasattr(m, attr):
            setattr(m, attr, use)
    return use
