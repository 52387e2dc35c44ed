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

__all__ = ["MaskPredictor", "PixelDecoder", "UniversalSegmentationHead"]


class MaskPredictor(nn.Module):
    def __init__(self, hidden_dim: int, mask_dim: int):
        super().__init__()
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)

    def forward(self, obj_queries: torch.Tensor, pixel_embed: torch.Tensor) -> torch.Tensor:
        """queries [(layers,) B, Q, C] x pixels [(B,) C, H, W] -> mask logits [(layers,) B, Q, H, W]."""
        q = self.mask_embed(obj_queries)
        lead = "l" if obj_queries.dim() == 4 else ""
        pix = "chw" if pixel_embed.ndim == 3 else "bchw"
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
            x = F.relu(self.norms[k](self.conv_layers[k](x)))
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
        levels.append(tokens.reshape(-1, *backbone_feats[-1].shape[1:]))
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
