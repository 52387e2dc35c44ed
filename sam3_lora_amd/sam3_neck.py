"""
Vision neck of the SAM3 image model and the vision/language backbone pair.

Restates ``sam3/model/position_encoding.py:10-124`` (``PositionEmbeddingSine``), ``sam3/model/necks.py:13-125``
(``Sam3DualViTDetNeck`` -- a SimpleFPN over the last ViT feature map) and ``sam3/model/vl_combiner.py:16-176``
(``SAM3VLBackbone``) for the training step of row a14, with the reference's module names
(``backbone.vision_backbone.{trunk,convs}``, ``backbone.language_backbone``).  PyTorch-ROCm ops only.

The sine table is a pure function of the feature-map size, so it is computed once per (size, device, dtype) and
expanded over the batch without a copy; the reference's constructor pre-fills its cache with a literal
``device="cuda"`` (position_encoding.py:47), which is not reproduced.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

__all__ = ["PositionEmbeddingSine", "Sam3DualViTDetNeck", "SAM3VLBackbone"]


def _interleaved_sincos(angle: torch.Tensor) -> torch.Tensor:
    """[..., F] angles -> [..., F] with sin on the even channels and cos on the odd ones (DETR convention)."""
    return torch.stack((angle[..., 0::2].sin(), angle[..., 1::2].cos()), dim=-1).flatten(-2)


class PositionEmbeddingSine(nn.Module):
    """2-D sine position code with ``num_pos_feats`` channels (half for y, half for x), coordinates normalised to
    (0, 2*pi]."""

    def __init__(self, num_pos_feats: int, temperature: int = 10000, normalize: bool = True,
                 scale: Optional[float] = None, precompute_resolution: Optional[int] = None):
        super().__init__()
        assert num_pos_feats % 2 == 0, "Expecting even model width"
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats // 2
        self.temperature, self.normalize = temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self.cache: Dict[Tuple, torch.Tensor] = {}

    def _freqs(self, device) -> torch.Tensor:
        k = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
        return self.temperature ** (2 * (k // 2) / self.num_pos_feats)

    def _encode_xy(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Normalised 1-D coordinate lists -> (code_x, code_y), each [N, num_pos_feats/2 * 2]."""
        assert x.ndim == y.ndim == 1 and len(x) == len(y)
        f = self._freqs(x.device)
        return _interleaved_sincos(x[:, None] * self.scale / f), _interleaved_sincos(y[:, None] * self.scale / f)

    @torch.no_grad()
    def encode_boxes(self, x, y, w, h) -> torch.Tensor:
        px, py = self._encode_xy(x, y)
        return torch.cat((py, px, h[:, None], w[:, None]), dim=1)

    encode = encode_boxes

    @torch.no_grad()
    def encode_points(self, x, y, labels) -> torch.Tensor:
        assert x.shape == y.shape == labels.shape
        px, py = self._encode_xy(x.flatten(), y.flatten())
        return torch.cat((py.reshape(*x.shape, -1), px.reshape(*x.shape, -1), labels[:, :, None]), dim=2)

    @torch.no_grad()
    def table(self, h: int, w: int, device) -> torch.Tensor:
        """[C, h, w] fp32."""
        key = (h, w, str(device))
        if key not in self.cache:
            ys = torch.arange(1, h + 1, dtype=torch.float32, device=device)
            xs = torch.arange(1, w + 1, dtype=torch.float32, device=device)
            if self.normalize:
                ys = ys / (ys[-1] + 1e-6) * self.scale
                xs = xs / (xs[-1] + 1e-6) * self.scale
            f = self._freqs(device)
            py = _interleaved_sincos(ys[:, None] / f)[:, None, :].expand(h, w, -1)
            px = _interleaved_sincos(xs[:, None] / f)[None, :, :].expand(h, w, -1)
            self.cache[key] = torch.cat((py, px), dim=2).permute(2, 0, 1).contiguous()
        return self.cache[key]

    @torch.no_grad()
    def forward(self, x: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """[B, C, h, w] as a stride-0 batch expansion of the cached table (cast, when asked, BEFORE the expansion: the
        cast of an expanded tensor would write B full copies -- 340 MB at the 288^2 level)."""
        t = self.table(x.shape[-2], x.shape[-1], x.device)
        if dtype is not None and dtype != t.dtype:
            key = (x.shape[-2], x.shape[-1], str(x.device), dtype)
            if key not in self.cache:
                self.cache[key] = t.to(dtype)
            t = self.cache[key]
        return t[None].expand(x.shape[0], -1, -1, -1)


class Sam3DualViTDetNeck(nn.Module):
    """SimpleFPN (ViTDet): from the trunk's single stride-14 map build one ``d_model``-channel map per scale factor
    (4: two stride-2 transposed convs with a GELU between, 2: one, 1: none, 0.5: 2x2 max-pool), each followed by a 1x1
    and a 3x3 conv.  ``forward(images) -> (features, position_codes, None, None)`` (the SAM2 twin neck is not built:
    ``enable_inst_interactivity`` is off in the image-training builder, model_builder.py:563)."""

    def __init__(self, trunk: nn.Module, position_encoding: nn.Module, d_model: int,
                 scale_factors: Sequence[float] = (4.0, 2.0, 1.0, 0.5), add_sam2_neck: bool = False):
        super().__init__()
        if add_sam2_neck:
            raise NotImplementedError("the SAM2 twin neck belongs to the interactive predictor, outside the training path")
        self.trunk, self.position_encoding = trunk, position_encoding
        self.scale_factors = scale_factors
        dim = trunk.channel_list[-1]
        self.convs = nn.ModuleList()
        for scale in scale_factors:
            stage = nn.Sequential()
            if scale == 4.0:
                stage.add_module("dconv_2x2_0", nn.ConvTranspose2d(dim, dim // 2, kernel_size=2, stride=2))
                stage.add_module("gelu", nn.GELU())
                stage.add_module("dconv_2x2_1", nn.ConvTranspose2d(dim // 2, dim // 4, kernel_size=2, stride=2))
                width = dim // 4
            elif scale == 2.0:
                stage.add_module("dconv_2x2", nn.ConvTranspose2d(dim, dim // 2, kernel_size=2, stride=2))
                width = dim // 2
            elif scale == 1.0:
                width = dim
            elif scale == 0.5:
                stage.add_module("maxpool_2x2", nn.MaxPool2d(kernel_size=2, stride=2))
                width = dim
            else:
                raise NotImplementedError(f"scale_factor={scale} is not supported yet.")
            stage.add_module("conv_1x1", nn.Conv2d(width, d_model, kernel_size=1, bias=True))
            stage.add_module("conv_3x3", nn.Conv2d(d_model, d_model, kernel_size=3, padding=1, bias=True))
            self.convs.append(stage)
        self.sam2_convs = None
        self.skip_coarsest = 0      # set by SAM3VLBackbone(scalp=...): levels nobody reads are not computed

    def forward(self, images: torch.Tensor):
        x = self.trunk(images)[-1]
        stages = list(self.convs)[:len(self.convs) - self.skip_coarsest]
        feats = [stage(x) for stage in stages]
        pos = [self.position_encoding(f, dtype=f.dtype) for f in feats]
        return feats, pos, None, None


class SAM3VLBackbone(nn.Module):
    """Vision neck + text tower side by side (no fusion).  ``scalp`` drops that many of the coarsest feature levels."""

    def __init__(self, visual: nn.Module, text: nn.Module, scalp: int = 0):
        super().__init__()
        self.vision_backbone = visual
        self.language_backbone = text
        self.scalp = scalp
        self._text_cache = {}
        if scalp > 0 and hasattr(visual, "skip_coarsest"):
            visual.skip_coarsest = scalp    # the reference computes the dropped levels and throws them away (:89-99)

    def forward_image(self, samples: torch.Tensor) -> Dict:
        feats, pos, _, _ = self.vision_backbone(samples)
        dropped = getattr(self.vision_backbone, "skip_coarsest", 0)
        if self.scalp > dropped:
            feats, pos = feats[:-(self.scalp - dropped)], pos[:-(self.scalp - dropped)]
        return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats, "sam2_backbone_out": None}

    def forward_text(self, captions: List[str], input_boxes=None, additional_text=None, device="cuda") -> Dict:
        texts = list(captions) + list(additional_text or [])
        mask, memory, embeds = self._encode_text(texts, input_boxes, device)
        out = {}
        if additional_text is not None:
            out["additional_text_features"] = memory[:, -len(additional_text):]
            out["additional_text_mask"] = mask[-len(additional_text):]
        n = len(captions)
        out["language_features"], out["language_mask"], out["language_embeds"] = memory[:, :n], mask[:n], embeds[:, :n]
        return out

    def _encode_text(self, texts, input_boxes, device):
        """The text tower is deterministic (no dropout) and, with no adapter inside it, frozen: its output for a given
        list of prompts is a constant of the run.  Cached per prompt list while nothing in the tower requires grad
        (keyed on the weights' version counters, so a load_state_dict refreshes it)."""
        tower = self.language_backbone
        if (not isinstance(texts[0], str)) or (input_boxes is not None and len(input_boxes) > 0) \
                or any(p.requires_grad for p in tower.parameters()):
            return tower(texts, input_boxes, device=device)
        w = tower.resizer.weight
        key = (tuple(texts), str(device), w.dtype, w._version, tower.encoder.token_embedding.weight._version)
        hit = self._text_cache.get(key)
        if hit is None:
            with torch.no_grad():
                hit = tower(texts, input_boxes, device=device)
            if len(self._text_cache) > 64:
                self._text_cache.clear()
            self._text_cache[key] = hit
        return hit

    def forward(self, samples, captions, input_boxes=None, additional_text=None):
        out = self.forward_image(samples)
        out.update(self.forward_text(captions, input_boxes, additional_text, out["vision_features"].device))
        return out
