"""
Mask head of the SAM3 image model ("_run_segmentation_heads"), on PyTorch-ROCm.

Restates ``sam3/model/maskformer_segmentation.py``: ``MaskPredictor`` :23-52, ``PixelDecoder`` :177-229 and
``UniversalSegmentationHead`` :232-323 (+ the parts of its base ``SegmentationHead`` :55-174 it uses) as
``model_builder.py:205-231`` configures them: the encoder's image tokens cross-attend to the prompt, replace the
coarsest FPN level, a three-stage nearest-upsampling FPN produces the pixel embedding at stride 3.5 (288 x 288 for a
1008 input), and every decoder query's mask is the dot product of its 3-layer-MLP embedding with the per-pixel
instance embedding.  Names are the reference's (``segmentation_head.*``; ``apply_to_mask_decoder`` gates on the
substring ``mask_decoder`` which does not occur in them -- a reference quirk the manifests in
``tests/golden/sam3_linears.json`` pin).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .sam3_detr import MLP

__all__ = ["MaskPredictor", "PixelDecoder", "UniversalSegmentationHead", "group_norm_relu", "gn_kernels_support"]


# ------------------------------------------------------------------ channels-last GroupNorm (+ ReLU), HIP kernels --
def gn_kernels_support(x: torch.Tensor, gn: nn.GroupNorm) -> bool:
    """The ``sam3_gn_nhwc_*`` kernels (include/sam3_seg_amd.h) apply: GPU map in bf16 / fp32, frozen affine parameters,
    a channel count the 16-byte vector geometry covers."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and gn.affine):
        return False
    if gn.weight.requires_grad or gn.bias.requires_grad or torch.is_autocast_enabled():
        return False
    from . import _ffi
    return _ffi.load().sam3_gn_nhwc_supported(x.shape[1], gn.num_groups, 0 if x.dtype == torch.bfloat16 else 1) == 0


def _seg_check(lib, rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib.sam3_seg_last_error().decode()}")


class _GroupNormNHWC(torch.autograd.Function):
    """``act(group_norm(x))`` on a channels-last map, frozen gamma / beta (fp32 copies); C-ABI ``sam3_gn_nhwc_fwd/bwd``.
    Input and output keep the channels-last memory the surrounding MIOpen convolutions use."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        x = x.contiguous(memory_format=torch.channels_last)
        N, C, H, W = x.shape
        dt = 0 if x.dtype == torch.bfloat16 else 1
        y = torch.empty_like(x)                                   # preserves the channels-last strides
        stats = torch.empty(N, groups, 2, device=x.device, dtype=torch.float32)
        ws = torch.empty(max(lib.sam3_gn_nhwc_workspace_bytes(N, H * W, C, groups), 16), device=x.device, dtype=torch.uint8)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _seg_check(lib, lib.sam3_gn_nhwc_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                             N, H * W, C, groups, eps, int(relu), dt, ws.data_ptr(), ws.numel(), st),
                   "sam3_gn_nhwc_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.meta = (groups, relu, dt)
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        x, gamma, beta, stats = ctx.saved_tensors
        groups, relu, dt = ctx.meta
        N, C, H, W = x.shape
        gy = gy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x)
        ws = torch.empty(max(lib.sam3_gn_nhwc_workspace_bytes(N, H * W, C, groups), 16), device=x.device, dtype=torch.uint8)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _seg_check(lib, lib.sam3_gn_nhwc_bwd(x.data_ptr(), gy.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             stats.data_ptr(), gx.data_ptr(), N, H * W, C, groups, int(relu), dt,
                                             ws.data_ptr(), ws.numel(), st), "sam3_gn_nhwc_bwd")
        return gx, None, None, None, None, None


def _affine_f32(gn: nn.GroupNorm):
    """fp32 copies of the frozen gamma / beta, refreshed when the parameters change (load_state_dict, .to())."""
    key = (gn.weight.data_ptr(), gn.weight._version, gn.bias.data_ptr(), gn.bias._version, gn.weight.device)
    held = gn.__dict__.get("_sam3_affine_f32")
    if held is None or held[0] != key:
        held = (key, gn.weight.detach().float().contiguous(), gn.bias.detach().float().contiguous())
        gn.__dict__["_sam3_affine_f32"] = held
    return held[1], held[2]


def group_norm_relu(x: torch.Tensor, gn: nn.GroupNorm, relu: bool = True) -> torch.Tensor:
    """``relu(gn(x))``: through the channels-last kernels where they apply, else the PyTorch operators."""
    if gn_kernels_support(x, gn):
        gamma, beta = _affine_f32(gn)
        return _GroupNormNHWC.apply(x, gamma, beta, gn.num_groups, float(gn.eps), relu)
    y = gn(x)
    return F.relu(y) if relu else y


class _MaskDot(torch.autograd.Function):
    """``einsum("...bqc,bchw->...bqhw")`` for a channels-last pixel embedding: the pixel gradient leaves as
    ``g^T q`` straight into channels-last memory (autograd's own formula writes it NCHW and the convolution before it
    converts 340 MB back)."""

    @staticmethod
    def forward(ctx, q, pix):
        B, C, H, W = pix.shape
        pv = pix.permute(0, 2, 3, 1).reshape(B, H * W, C)                  # a view of the channels-last memory
        lead = q.shape[:-3]
        q3 = q.reshape(-1, *q.shape[-3:]) if lead else q.unsqueeze(0)      # [L, B, Q, C]
        L, _, Q, _ = q3.shape
        qb = q3.permute(1, 0, 2, 3).reshape(B, L * Q, C)
        out = torch.bmm(qb, pv.transpose(1, 2))                            # [B, L*Q, HW]
        ctx.save_for_backward(qb, pv)
        ctx.dims = (L, Q, B, C, H, W, tuple(q.shape))
        out = out.reshape(B, L, Q, H, W).permute(1, 0, 2, 3, 4)
        return out.reshape(*q.shape[:-1], H, W) if L > 1 or lead else out[0]

    @staticmethod
    def backward(ctx, g):
        qb, pv = ctx.saved_tensors
        L, Q, B, C, H, W, qshape = ctx.dims
        g3 = g.reshape(L, B, Q, H * W).permute(1, 0, 2, 3).reshape(B, L * Q, H * W)
        gq = gpix = None
        if ctx.needs_input_grad[0]:
            gq = torch.bmm(g3, pv).reshape(B, L, Q, C).permute(1, 0, 2, 3).reshape(qshape)
        if ctx.needs_input_grad[1]:
            gp = torch.bmm(g3.transpose(1, 2), qb)                         # [B, HW, C]: channels-last memory
            gpix = gp.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return gq, gpix


class MaskPredictor(nn.Module):
    def __init__(self, hidden_dim: int, mask_dim: int):
        super().__init__()
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)

    def forward(self, obj_queries: torch.Tensor, pixel_embed: torch.Tensor) -> torch.Tensor:
        """queries [(layers,) B, Q, C] x pixels [(B,) C, H, W] -> mask logits [(layers,) B, Q, H, W]."""
        q = self.mask_embed(obj_queries)
        lead = "l" if obj_queries.dim() == 4 else ""
        pix = "chw" if pixel_embed.ndim == 3 else "bchw"
        if (pixel_embed.ndim == 4 and pixel_embed.is_cuda and not pixel_embed.is_contiguous()
                and pixel_embed.is_contiguous(memory_format=torch.channels_last) and q.dtype == pixel_embed.dtype
                and q.shape[-3] == pixel_embed.shape[0]):
            return _MaskDot.apply(q, pixel_embed)
        return torch.einsum(f"{lead}bqc,{pix}->{lead}bqhw", q, pixel_embed)


class PixelDecoder(nn.Module):
    """Top-down FPN: starting from the coarsest map, repeatedly upsample (nearest) to the next finer level, add it,
    3x3 conv, GroupNorm(8), ReLU."""

    def __init__(self, hidden_dim: int, num_upsampling_stages: int, interpolation_mode: str = "nearest",
                 shared_conv: bool = False):
        super().__init__()
        self.hidden_dim, self.num_upsampling_stages = hidden_dim, num_upsampling_stages
        self.interpolation_mode, self.shared_conv = interpolation_mode, shared_conv
        n = 1 if shared_conv else num_upsampling_stages
        self.conv_layers = nn.ModuleList(nn.Conv2d(hidden_dim, hidden_dim, 3, 1, 1) for _ in range(n))
        self.norms = nn.ModuleList(nn.GroupNorm(8, hidden_dim) for _ in range(n))
        self.out_dim = hidden_dim

    def forward(self, backbone_feats: List[torch.Tensor]) -> torch.Tensor:
        x = backbone_feats[-1]
        for i, finer in enumerate(reversed(backbone_feats[:-1])):
            k = 0 if self.shared_conv else i
            x = finer + F.interpolate(x, size=finer.shape[-2:], mode=self.interpolation_mode)
            x = group_norm_relu(self.conv_layers[k](x), self.norms[k])
        return x


class UniversalSegmentationHead(nn.Module):
    """Instance + semantic mask head.  ``forward(...) -> {"pred_masks", "semantic_seg", "presence_logit"}``."""

    def __init__(self, hidden_dim: int, upsampling_stages: int, pixel_decoder: nn.Module, aux_masks: bool = False,
                 no_dec: bool = False, act_ckpt: bool = False, presence_head: bool = False, dot_product_scorer=None,
                 cross_attend_prompt: Optional[nn.Module] = None):
        super().__init__()
        assert not no_dec and not presence_head and dot_product_scorer is None, "not part of the SAM3 image builder"
        self.use_encoder_inputs, self.aux_masks, self.no_dec, self.act_ckpt = True, aux_masks, no_dec, act_ckpt
        self.pixel_decoder = pixel_decoder
        self.mask_predictor = MaskPredictor(hidden_dim, mask_dim=hidden_dim)
        self.instance_keys = ["pred_masks"]
        self.d_model = hidden_dim
        self.presence_head = None
        self.cross_attend_prompt = cross_attend_prompt
        if cross_attend_prompt is not None:
            self.cross_attn_norm = nn.LayerNorm(hidden_dim)
        self.semantic_seg_head = nn.Conv2d(pixel_decoder.out_dim, 1, kernel_size=1)
        self.instance_seg_head = nn.Conv2d(pixel_decoder.out_dim, hidden_dim, kernel_size=1)

    def _embed_pixels(self, backbone_feats: List[torch.Tensor], image_ids, encoder_hidden_states: torch.Tensor):
        """FPN levels (per image when the batch has several prompts per image) with the coarsest level replaced by
        the encoder's output tokens reshaped to its map."""
        finer = backbone_feats[:-1]                  # the coarsest level is replaced below: never gathered
        if image_ids is None:                        # caller knows prompt i reads image i (sam3_data.collate_fn_api)
            levels = list(finer)
        elif backbone_feats[0].shape[0] > 1:
            levels = [f[image_ids.to(f.device), ...] for f in finer]
        else:                                        # one image: broadcasting does the per-prompt copy
            levels = list(finer)
        hw = math.prod(backbone_feats[-1].shape[-2:])
        tokens = encoder_hidden_states.permute(1, 2, 0)[..., :hw]
        coarse = tokens.reshape(-1, *backbone_feats[-1].shape[1:])
        if coarse.is_cuda:      # [HW, B, C] memory seen as [B, C, H, W]: make it the channels-last map the FPN levels are
            coarse = coarse.contiguous(memory_format=torch.channels_last)
        levels.append(coarse)
        if self.act_ckpt and torch.is_grad_enabled():
            return checkpoint(self.pixel_decoder, levels, use_reentrant=False)
        return self.pixel_decoder(levels)

    def forward(self, backbone_feats: List[torch.Tensor], obj_queries: torch.Tensor, image_ids,
                encoder_hidden_states: Optional[torch.Tensor] = None, prompt: Optional[torch.Tensor] = None,
                prompt_mask: Optional[torch.Tensor] = None, **kwargs) -> Dict[str, Optional[torch.Tensor]]:
        assert encoder_hidden_states is not None
        if self.cross_attend_prompt is not None:
            h = self.cross_attend_prompt(query=self.cross_attn_norm(encoder_hidden_states), key=prompt, value=prompt,
                                         key_padding_mask=prompt_mask)[0]
            encoder_hidden_states = h + encoder_hidden_states
        pixel_embed = self._embed_pixels(backbone_feats, image_ids, encoder_hidden_states)
        instance_embed = self.instance_seg_head(pixel_embed)
        queries = obj_queries if self.aux_masks else obj_queries[-1]
        return {"pred_masks": self.mask_predictor(queries, instance_embed),
                "semantic_seg": self.semantic_seg_head(pixel_embed), "presence_logit": None}
