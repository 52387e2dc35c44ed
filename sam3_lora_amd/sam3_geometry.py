"""
Geometric-prompt encoder of the SAM3 image model ("_encode_prompt").

Restates ``sam3/model/geometry_encoders.py``: ``concat_padded_sequences`` :22-79, ``Prompt`` :82-412 and
``SequenceGeometryEncoder`` :481-850 in the configuration ``model_builder.py:234-290`` builds (boxes and points
each encoded as direct projection + pooled image feature + position code, a CLS token, a projection + norm, then
three transformer layers cross-attending to the image).  In LoRA training the prompt is text only, so the sequence
the encoder sees is empty and its output is the encoded CLS token; box and point prompts are supported for the
consumers of the trained adapters.  Box pooling is torchvision's RoIAlign in the reference
(geometry_encoders.py:662-664, ``aligned=False``, adaptive sampling); torchvision is not a dependency here, so the
same sampling rule is written out in :func:`roi_align`.

Module and parameter names are the reference's (``geometry_encoder.*``: the ``apply_to_geometry_encoder`` gate and the
``*_direct_project`` / ``*_pool_project`` adapter targets of the package API).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .matcher import box_cxcywh_to_xyxy
from .sam3_detr import clones

__all__ = ["Prompt", "SequenceGeometryEncoder", "concat_padded_sequences", "roi_align"]


def concat_padded_sequences(seq1, mask1, seq2, mask2, return_index: bool = False):
    """Append right-padded ``seq2[L2, B, C]`` after the valid part of right-padded ``seq1[L1, B, C]`` per batch item
    (masks ``[B, L]``, True = padding).  Result ``[L1 + L2, B, C]`` is right-padded again."""
    L1, B, C = seq1.shape
    L2 = seq2.shape[0]
    assert seq2.shape[1:] == (B, C) and mask1.shape == (B, L1) and mask2.shape == (B, L2)
    len1 = (~mask1).sum(dim=-1)
    total = len1 + (~mask2).sum(dim=-1)
    out_mask = torch.arange(L1 + L2, device=seq2.device)[None].repeat(B, 1) >= total[:, None]
    out = torch.zeros((L1 + L2, B, C), device=seq2.device, dtype=seq2.dtype)
    out[:L1] = seq1
    where = torch.arange(L2, device=seq2.device)[:, None].repeat(1, B) + len1[None]
    out = out.scatter(0, where[:, :, None].expand(-1, -1, C), seq2)
    return (out, out_mask, where) if return_index else (out, out_mask)


class Prompt:
    """Box / point (/ mask) prompts, sequence-first: ``box_embeddings [Nb, B, 4]`` cxcywh, ``box_mask [B, Nb]``
    (True = padding), ``box_labels [Nb, B]``; points likewise with 2 coordinates.  Missing parts default to empty
    sequences with positive labels."""

    def __init__(self, box_embeddings=None, box_mask=None, point_embeddings=None, point_mask=None, box_labels=None,
                 point_labels=None, mask_embeddings=None, mask_mask=None, mask_labels=None):
        given = [t for t in (box_embeddings, point_embeddings, mask_embeddings) if t is not None]
        self.mask_embeddings, self.mask_mask, self.mask_labels = mask_embeddings, mask_mask, mask_labels
        if not given:
            self.box_embeddings = self.box_labels = self.box_mask = None
            self.point_embeddings = self.point_labels = self.point_mask = None
            return
        B, dev = given[0].shape[1], given[0].device
        assert all(t.shape[1] == B and t.device == dev for t in given), "prompt parts disagree on batch size / device"

        def fill(emb, labels, mask, width):
            n = emb.shape[0] if emb is not None else 0
            emb = torch.zeros(n, B, width, device=dev) if emb is None else emb
            labels = torch.ones(n, B, device=dev, dtype=torch.long) if labels is None else labels
            mask = torch.zeros(B, n, device=dev, dtype=torch.bool) if mask is None else mask
            assert list(emb.shape[:2]) == [n, B] and list(mask.shape) == [B, n] and list(labels.shape) == [n, B]
            return emb, labels, mask

        self.box_embeddings, self.box_labels, self.box_mask = fill(box_embeddings, box_labels, box_mask, 4)
        self.point_embeddings, self.point_labels, self.point_mask = fill(point_embeddings, point_labels, point_mask, 2)
        if mask_embeddings is not None:
            n = mask_embeddings.shape[0]
            if mask_labels is None:
                self.mask_labels = torch.ones(n, B, device=dev, dtype=torch.long)
            if mask_mask is None:
                self.mask_mask = torch.zeros(B, n, device=dev, dtype=torch.bool)

    def _append(self, kind: str, emb, labels, mask):
        cur = getattr(self, f"{kind}_embeddings")
        if cur is None:
            setattr(self, f"{kind}_embeddings", emb)
            setattr(self, f"{kind}_labels", labels)
            setattr(self, f"{kind}_mask", mask)
            return
        B = cur.shape[1]
        assert emb.shape[1] == labels.shape[1] == B and list(emb.shape[:2]) == list(labels.shape[:2])
        if mask is None:
            mask = torch.zeros(B, emb.shape[0], dtype=torch.bool, device=emb.device)
        cur_mask = getattr(self, f"{kind}_mask")
        lab, _ = concat_padded_sequences(getattr(self, f"{kind}_labels").unsqueeze(-1), cur_mask, labels.unsqueeze(-1), mask)
        new, new_mask = concat_padded_sequences(cur, cur_mask, emb, mask)
        setattr(self, f"{kind}_embeddings", new)
        setattr(self, f"{kind}_labels", lab.squeeze(-1))
        setattr(self, f"{kind}_mask", new_mask)

    def append_boxes(self, boxes, labels, mask=None):
        self._append("box", boxes, labels, mask)

    def append_points(self, points, labels, mask=None):
        self._append("point", points, labels, mask)

    def clone(self) -> "Prompt":
        c = lambda t: None if t is None else t.clone()
        return Prompt(box_embeddings=c(self.box_embeddings), box_mask=c(self.box_mask),
                      point_embeddings=c(self.point_embeddings), point_mask=c(self.point_mask),
                      box_labels=c(self.box_labels), point_labels=c(self.point_labels))


def roi_align(feats: torch.Tensor, boxes: Sequence[torch.Tensor], output_size: int) -> torch.Tensor:
    """RoIAlign with torchvision's defaults (spatial_scale 1, adaptive sampling ``ceil(roi / output_size)`` points per
    bin and axis, ``aligned=False``: no half-pixel shift, ROI extent at least 1).  ``feats [B, C, H, W]``, ``boxes``:
    one ``[n_i, 4]`` xyxy tensor (feature-map pixels) per image -> ``[sum n_i, C, S, S]``."""
    B, C, H, W = feats.shape
    S = output_size
    out = []
    for b, per_image in enumerate(boxes):
        if per_image.shape[0] == 0:         # text-only prompts (the training path): nothing to read back from the device
            continue
        fm = feats[b]
        for box in per_image.tolist():
            x1, y1, x2, y2 = box
            rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
            gw, gh = max(int(math.ceil(rw / S)), 1), max(int(math.ceil(rh / S)), 1)
            ys = y1 + (torch.arange(S * gh, device=feats.device, dtype=torch.float32) + 0.5) * (rh / (S * gh))
            xs = x1 + (torch.arange(S * gw, device=feats.device, dtype=torch.float32) + 0.5) * (rw / (S * gw))

            def axis(v, n):             # -> (low index, high index, high weight, inside flag)
                inside = (v >= -1.0) & (v <= n)
                v = v.clamp(min=0)
                lo = v.floor().long()
                top = lo >= n - 1
                lo = torch.where(top, torch.full_like(lo, n - 1), lo)
                hi = torch.where(top, lo, lo + 1)
                w_hi = torch.where(top, torch.zeros_like(v), v - lo.float())
                return lo, hi, w_hi, inside

            ylo, yhi, wy, iny = axis(ys, H)
            xlo, xhi, wx, inx = axis(xs, W)
            f = fm.float()
            top = f[:, ylo][:, :, xlo] * (1 - wx) + f[:, ylo][:, :, xhi] * wx
            bot = f[:, yhi][:, :, xlo] * (1 - wx) + f[:, yhi][:, :, xhi] * wx
            val = top * (1 - wy)[None, :, None] + bot * wy[None, :, None]
            val = val * (iny[:, None] & inx[None, :])[None]
            out.append(val.reshape(C, S, gh, S, gw).mean(dim=(2, 4)).to(feats.dtype))
    if not out:
        return feats.new_zeros((0, C, S, S))
    return torch.stack(out)


class SequenceGeometryEncoder(nn.Module):
    def __init__(self, encode_boxes_as_points: bool, points_direct_project: bool, points_pool: bool,
                 points_pos_enc: bool, boxes_direct_project: bool, boxes_pool: bool, boxes_pos_enc: bool, d_model: int,
                 pos_enc, num_layers: int, layer: nn.Module, roi_size: int = 7, add_cls: bool = True,
                 add_post_encode_proj: bool = True, mask_encoder=None, add_mask_label: bool = False,
                 use_act_ckpt: bool = False):
        super().__init__()
        assert mask_encoder is None, "mask prompts belong to the interactive / video models, outside the path"
        assert points_direct_project or points_pos_enc or points_pool, "Error: need at least one way to encode points"
        assert encode_boxes_as_points or boxes_direct_project or boxes_pos_enc or boxes_pool, \
            "Error: need at least one way to encode boxes"
        self.d_model, self.pos_enc, self.roi_size = d_model, pos_enc, roi_size
        self.encode_boxes_as_points = encode_boxes_as_points
        self.label_embed = nn.Embedding(6 if encode_boxes_as_points else 2, d_model)
        self.cls_embed = nn.Embedding(1, d_model) if add_cls else None
        self.points_direct_project = nn.Linear(2, d_model) if points_direct_project else None
        self.points_pool_project = nn.Linear(d_model, d_model) if points_pool else None
        self.points_pos_enc_project = nn.Linear(d_model, d_model) if points_pos_enc else None
        self.boxes_direct_project = self.boxes_pool_project = self.boxes_pos_enc_project = None
        if not encode_boxes_as_points:
            if boxes_direct_project:
                self.boxes_direct_project = nn.Linear(4, d_model)
            if boxes_pool:
                self.boxes_pool_project = nn.Conv2d(d_model, d_model, roi_size)
            if boxes_pos_enc:
                self.boxes_pos_enc_project = nn.Linear(d_model + 2, d_model)
        self.final_proj = None
        if add_post_encode_proj:
            self.final_proj = nn.Linear(d_model, d_model)
            self.norm = nn.LayerNorm(d_model)
        pools = self.points_pool_project is not None or self.boxes_pool_project is not None
        self.img_pre_norm = nn.LayerNorm(d_model) if pools else nn.Identity()
        self.encode = None
        if num_layers > 0:
            assert add_cls, "It's currently highly recommended to add a CLS when using a transformer"
            self.encode = clones(layer, num_layers)
            self.encode_norm = nn.LayerNorm(d_model)
        self.add_mask_label, self.mask_encoder, self.use_act_ckpt = add_mask_label, None, use_act_ckpt

    # -- one embedding per point / box: the sum of the enabled encodings plus the label embedding --------------
    def _encode_points(self, points, labels, img_nchw):
        """Coordinates stay fp32 for the geometry; what enters a Linear takes the layer's dtype (bf16 layout)."""
        n, bs = points.shape[:2]
        wd = self.label_embed.weight.dtype
        points = points.float()
        parts = []
        if self.points_direct_project is not None:
            parts.append(self.points_direct_project(points.to(wd)))
        if self.points_pool_project is not None:
            grid = points.transpose(0, 1).unsqueeze(2) * 2 - 1               # [B, n, 1, 2] in [-1, 1]
            picked = F.grid_sample(img_nchw, grid.to(img_nchw.dtype), align_corners=False)      # [B, C, n, 1]
            parts.append(self.points_pool_project(picked.squeeze(-1).permute(2, 0, 1)))
        if self.points_pos_enc_project is not None:
            x, y = points.unbind(-1)
            ex, ey = self.pos_enc._encode_xy(x.flatten(), y.flatten())
            code = torch.cat([ex.view(n, bs, ex.shape[-1]), ey.view(n, bs, ey.shape[-1])], -1)
            parts.append(self.points_pos_enc_project(code.to(wd)))
        return self.label_embed(labels.long()) + sum(parts[1:], parts[0])

    def _encode_boxes(self, boxes, labels, img_nchw):
        n, bs = boxes.shape[:2]
        wd = self.label_embed.weight.dtype
        boxes = boxes.float()
        parts = []
        if self.boxes_direct_project is not None:
            parts.append(self.boxes_direct_project(boxes.to(wd)))
        if self.boxes_pool_project is not None:
            H, W = img_nchw.shape[-2:]
            xyxy = box_cxcywh_to_xyxy(boxes)        # to feature-map pixels; python scalars, no host -> device copy
            px = torch.stack((xyxy[..., 0] * W, xyxy[..., 1] * H, xyxy[..., 2] * W, xyxy[..., 3] * H), dim=-1)
            pooled = roi_align(img_nchw, px.float().transpose(0, 1).unbind(0), self.roi_size)
            parts.append(self.boxes_pool_project(pooled).view(bs, n, self.d_model).transpose(0, 1))
        if self.boxes_pos_enc_project is not None:
            cx, cy, w, h = boxes.unbind(-1)
            code = self.pos_enc.encode_boxes(cx.flatten(), cy.flatten(), w.flatten(), h.flatten())
            parts.append(self.boxes_pos_enc_project(code.view(n, bs, code.shape[-1]).to(wd)))
        return self.label_embed(labels.long()) + sum(parts[1:], parts[0])

    def forward(self, geo_prompt: Prompt, img_feats: List[torch.Tensor], img_sizes, img_pos_embeds=None):
        """``img_feats`` / ``img_pos_embeds``: per level ``[HW, B, C]``.  Returns ``(embeddings [L, B, C], mask [B, L])``."""
        points, points_mask, points_labels = geo_prompt.point_embeddings, geo_prompt.point_mask, geo_prompt.point_labels
        boxes, boxes_mask, boxes_labels = geo_prompt.box_embeddings, geo_prompt.box_mask, geo_prompt.box_labels
        memory = img_feats[-1]
        memory_pos = img_pos_embeds[-1] if img_pos_embeds is not None else torch.zeros_like(memory)
        img_nchw = None
        if self.points_pool_project is not None or self.boxes_pool_project is not None:
            assert len(img_feats) == len(img_sizes)
            H, W = img_sizes[-1]
            normed = self.img_pre_norm(memory)
            assert normed.shape[0] == H * W
            img_nchw = normed.permute(1, 2, 0).reshape(normed.shape[1], normed.shape[2], H, W)

        if self.encode_boxes_as_points:        # each box becomes its two corner points with their own label range
            xyxy = box_cxcywh_to_xyxy(boxes)
            for corner, shift in ((xyxy[..., :2], 2), (xyxy[..., 2:], 4)):
                lab, _ = concat_padded_sequences(points_labels.unsqueeze(-1), points_mask,
                                                 (boxes_labels + shift).unsqueeze(-1), boxes_mask)
                points, points_mask = concat_padded_sequences(points, points_mask, corner, boxes_mask)
                points_labels = lab.squeeze(-1)
        seq, seq_mask = self._encode_points(points, points_labels, img_nchw), points_mask
        if not self.encode_boxes_as_points:
            seq, seq_mask = concat_padded_sequences(seq, seq_mask, self._encode_boxes(boxes, boxes_labels, img_nchw),
                                                    boxes_mask)
        bs = seq.shape[1]
        if self.cls_embed is not None:
            cls = self.cls_embed.weight.view(1, 1, self.d_model).repeat(1, bs, 1)
            seq, seq_mask = concat_padded_sequences(seq, seq_mask, cls,
                                                    torch.zeros(bs, 1, dtype=seq_mask.dtype, device=seq_mask.device))
        if self.final_proj is not None:
            seq = self.norm(self.final_proj(seq))
        if self.encode is not None:
            ckpt = self.training and self.use_act_ckpt and torch.is_grad_enabled()
            for lay in self.encode:
                def run(t, m, kpm, p, lay=lay):
                    return lay(t, m, tgt_key_padding_mask=kpm, pos=p)
                seq = checkpoint(run, seq, memory, seq_mask, memory_pos, use_reentrant=False) if ckpt \
                    else run(seq, memory, seq_mask, memory_pos)
            seq = self.encode_norm(seq)
        return seq, seq_mask
