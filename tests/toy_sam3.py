"""
Stand-in for the two plug-in points of sam3_lora_amd.trainer: a small detector built on the library's ViT trunk
that emits the SAM3 output dictionary (pred_logits / pred_boxes / pred_boxes_xyxy / pred_masks /
presence_logit_dec / aux_outputs) and a synthetic "one square per image" dataset with the target dictionary
``back_convert`` produces.  Module names follow the real model (``backbone.vision_backbone.trunk``,
``transformer.decoder``) so that the ``apply_to_*`` gates of LoRAConfig are exercised.
"""
from dataclasses import dataclass
from typing import Dict, List

import torch
from torch import nn

from sam3_lora_amd.matcher import box_cxcywh_to_xyxy
from sam3_lora_amd.vit import ViT

IMG = 56


@dataclass
class ToyBatch:
    img_batch: torch.Tensor
    find_targets: List[Dict]


class _Decoder(nn.Module):
    def __init__(self, dim, queries):
        super().__init__()
        self.query = nn.Parameter(torch.randn(queries, dim) * 0.5)
        self.q_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.box = nn.Linear(dim, 4)
        self.cls = nn.Linear(dim, 1)
        self.presence = nn.Linear(dim, 1)

    def forward(self, feat):                           # feat [B, C, h, w]
        B, C, h, w = feat.shape
        tok = feat.flatten(2).transpose(1, 2).to(self.query.dtype)          # [B, hw, C]
        q = self.q_proj(self.query)[None].expand(B, -1, -1)
        att = torch.softmax(q @ tok.transpose(1, 2) / C ** 0.5, dim=-1)
        outs = []
        hs = q
        for _ in range(2):                              # two "layers": first is the aux output
            hs = hs + att @ self.v_proj(tok)
            boxes = self.box(hs).float().sigmoid()
            outs.append({"pred_logits": self.cls(hs).float(), "pred_boxes": boxes, "pred_boxes_xyxy": box_cxcywh_to_xyxy(boxes),
                         "presence_logit_dec": self.presence(hs.mean(1)).float()})
        final = outs[-1]
        final["pred_masks"] = (hs @ tok.transpose(1, 2)).float().reshape(B, -1, h, w)
        final["aux_outputs"] = outs[:-1]
        return final


class ToySam3(nn.Module):
    def __init__(self, queries=5, dim=64):
        super().__init__()
        self.backbone = nn.Module()
        self.backbone.vision_backbone = nn.Module()
        self.backbone.vision_backbone.trunk = ViT(img_size=IMG, pretrain_img_size=IMG, patch_size=14, embed_dim=dim,
                                                  depth=2, num_heads=2, mlp_ratio=4.0, drop_path_rate=0.0, window_size=2,
                                                  global_att_blocks=(1,), use_act_checkpoint=True)
        self.transformer = nn.Module()
        self.transformer.decoder = _Decoder(dim, queries)

    def forward(self, batch: ToyBatch):
        x = batch.img_batch
        trunk = self.backbone.vision_backbone.trunk
        feat = trunk(x.to(trunk.patch_embed.proj.weight.dtype))[-1]         # the trunk's dtype (the decoder may be an fp32 island)
        return [self.transformer.decoder(feat)]            # one find stage

    @staticmethod
    def back_convert(t):
        return t


def model_builder(config, device):
    torch.manual_seed(int(config.get("seed", 0)))
    return ToySam3().to(device)


class _Loader:
    def __init__(self, n_batches, batch, seed):
        self.n, self.b, self.seed = n_batches, batch, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            cx = torch.rand(self.b, 2, generator=g) * 0.5 + 0.25
            wh = torch.rand(self.b, 2, generator=g) * 0.2 + 0.2
            boxes = torch.cat([cx, wh], -1)
            xyxy = box_cxcywh_to_xyxy(boxes)
            ys = (torch.arange(IMG) + 0.5) / IMG
            inside = lambda lo, hi: (ys[None] >= lo[:, None]) & (ys[None] < hi[:, None])
            masks = inside(xyxy[:, 1], xyxy[:, 3])[:, :, None] & inside(xyxy[:, 0], xyxy[:, 2])[:, None, :]
            img = masks[:, None].float().expand(-1, 3, -1, -1) + 0.05 * torch.randn(self.b, 3, IMG, IMG, generator=g)
            tgt = dict(boxes=boxes, boxes_xyxy=xyxy, boxes_padded=boxes[:, None], num_boxes=torch.ones(self.b, dtype=torch.long),
                       object_ids_padded=torch.zeros(self.b, 1, dtype=torch.long),
                       is_exhaustive=torch.ones(self.b, dtype=torch.bool), masks=masks,
                       is_valid_mask=torch.ones(self.b, dtype=torch.bool))
            yield {"input": ToyBatch(img_batch=img, find_targets=[tgt])}


def data_builder(config, split):
    rank = int(__import__("os").environ.get("RANK", "0"))
    bs = config["training"]["batch_size"]
    if split == "train":
        return _Loader(int(config.get("toy_train_batches", 6)), bs, seed=100 + rank)
    if config.get("toy_no_valid"):
        raise FileNotFoundError("no valid split")
    return _Loader(2, bs, seed=999)
