"""
Consumers of the trained adapters: inference and validation with ``*_lora_weights.pt``, and
adapter-free export.

Restates the parts of the reference's post-training scripts that touch the adapter files:
  * ``infer_sam.py:59-352`` -- ``SAM3LoRAInference``: config + weights -> model in eval mode, one text prompt per
    forward, sigmoid scores above a threshold, boxes scaled to the original image, masks resized to it;
  * ``validate_sam3_lora.py:232-350,770-1090`` -- mask NMS with a score pre-filter (``sam3/perflib/nms.py:24-88``:
    greedy, on the pairwise mask-IoU matrix), optional merging of overlapping segments, and metrics over a COCO split.
    The reference computes mAP / cgF1 with pycocotools and its own evaluators (not installed here, ~8k lines outside
    the path); this module computes mask AP with COCO's definition (101-point interpolated precision, IoU thresholds
    0.50:0.05:0.95, detections ranked over the whole split) plus precision / recall / F1 at IoU 0.5 -- stated as such
    in the printed report;
  * adapter-free export: ``merge_lora_into_linear`` folds every adapter into its frozen Linear
    (``W + (alpha/r) (A B)^T`` through the C-ABI ``sam3_lora_merge``; the package API's ``merge_lora_weights`` is the
    reference for it, ``sam3_lora/lora/lora_utils.py:230-255``) so that inference needs no adapter code at all.

The adapter arithmetic inside the model runs on the HIP path as in training; everything in this file is host logic.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

from .lora_layers import LoRAConfig, LoRALinear, apply_lora_to_model, count_parameters, load_lora_weights
from .sam3_data import (COCOSegmentDataset, Datapoint, FindQueryLoaded, Image, InferenceMetadata, collate_fn_api)
from .trainer import LORA_KEYS, default_model_builder, move_to_device

__all__ = ["SAM3LoRAInference", "nms_masks", "apply_sam3_nms", "merge_overlapping_masks", "mask_average_precision",
           "validate", "merge_lora_into_linear", "build_adapted_model"]


# ------------------------------------------------------------------------------------------- post-processing --
def _pairwise_mask_iou(masks: torch.Tensor) -> torch.Tensor:
    flat = masks.flatten(1).float()
    inter = flat @ flat.t()
    area = flat.sum(1)
    union = area[:, None] + area[None, :] - inter
    return inter / union.clamp(min=1.0)


def nms_masks(pred_probs: torch.Tensor, pred_masks: torch.Tensor, prob_threshold: float, iou_threshold: float) -> torch.Tensor:
    """Keep-flags [N]: detections above ``prob_threshold``, then greedy NMS by descending score on mask IoU
    (a detection suppresses every later one whose IoU with it exceeds ``iou_threshold``)."""
    valid = pred_probs > prob_threshold
    keep = torch.zeros_like(valid)
    idx = torch.nonzero(valid).squeeze(1)
    if idx.numel() == 0:
        return keep
    iou = _pairwise_mask_iou(pred_masks[idx] > 0).cpu().numpy()
    order = np.argsort(-pred_probs[idx].float().cpu().numpy(), kind="stable")
    alive = np.ones(len(order), dtype=bool)
    for pos, i in enumerate(order):
        if not alive[pos]:
            continue
        keep[idx[i]] = True
        rest = order[pos + 1:]
        alive[pos + 1:] &= iou[i, rest] <= iou_threshold
    return keep


def apply_sam3_nms(pred_logits, pred_masks, pred_boxes, prob_threshold: float = 0.3, nms_iou_threshold: float = 0.7,
                   max_detections: int = 100):
    """(mask probabilities, scores, boxes) of the detections that survive the score filter, mask NMS and top-k."""
    if len(pred_logits) == 0:
        return pred_masks[:0], pred_logits[:0].squeeze(-1), pred_boxes[:0]
    probs = torch.sigmoid(pred_logits.float()).squeeze(-1)
    mask_probs = torch.sigmoid(pred_masks.float())
    keep = nms_masks(probs, (mask_probs > 0.5).float(), prob_threshold, nms_iou_threshold)
    mask_probs, probs, boxes = mask_probs[keep], probs[keep], pred_boxes[keep]
    if 0 < max_detections < len(probs):
        probs, top = torch.topk(probs, k=max_detections)
        mask_probs, boxes = mask_probs[top], boxes[top]
    return mask_probs, probs, boxes


def merge_overlapping_masks(binary_masks, scores, boxes, iou_threshold: float = 0.15):
    """Union detections that overlap (IoU above the threshold with the growing union), best score first: the
    crack-segments heuristic of the reference's validation script."""
    if len(binary_masks) == 0:
        return binary_masks, scores, boxes
    order = torch.argsort(scores, descending=True)
    binary_masks, scores, boxes = binary_masks[order], scores[order], boxes[order]
    used = [False] * len(binary_masks)
    out_m, out_s, out_b = [], [], []
    for i in range(len(binary_masks)):
        if used[i]:
            continue
        cur, best = binary_masks[i].clone(), scores[i].item()
        used[i] = True
        for j in range(i + 1, len(binary_masks)):
            if used[j]:
                continue
            union = (cur | binary_masks[j]).sum().item()
            if union > 0 and (cur & binary_masks[j]).sum().item() / union > iou_threshold:
                cur |= binary_masks[j]
                best = max(best, scores[j].item())
                used[j] = True
        out_m.append(cur), out_s.append(best), out_b.append(boxes[i])
    return torch.stack(out_m), torch.tensor(out_s, device=scores.device), torch.stack(out_b)


def mask_average_precision(detections: Sequence[Tuple[int, float, np.ndarray]], ground_truth: Dict[int, List[np.ndarray]],
                           iou_thresholds: Sequence[float] = tuple(np.arange(0.5, 0.96, 0.05))) -> Dict[str, float]:
    """COCO-style mask AP for one category.  ``detections``: (image id, score, bool mask); ``ground_truth``: image id
    -> list of bool masks.  Per IoU threshold: detections ranked by score over the whole split, each matched greedily
    to the unmatched ground truth of its image with the highest IoU above the threshold; AP = mean of the interpolated
    precision at 101 recall points.  Also precision / recall / F1 at IoU 0.5 over all detections given."""
    n_gt = sum(len(v) for v in ground_truth.values())
    dets = sorted(detections, key=lambda d: -d[1])
    ious = []
    for img, _, m in dets:
        gts = ground_truth.get(img, [])
        if not gts:
            ious.append(np.zeros(0))
            continue
        mf = m.reshape(-1).astype(np.float32)
        g = np.stack([x.reshape(-1).astype(np.float32) for x in gts])
        inter = g @ mf
        union = g.sum(1) + mf.sum() - inter
        ious.append(inter / np.maximum(union, 1.0))
    aps = {}
    recall_points = np.linspace(0, 1, 101)
    tp50 = 0
    for thr in (round(float(t), 2) for t in iou_thresholds):
        taken = {img: np.zeros(len(g), dtype=bool) for img, g in ground_truth.items()}
        tp = np.zeros(len(dets))
        for k, (img, _, _) in enumerate(dets):
            iou = ious[k]
            if iou.size == 0:
                continue
            cand = np.where(~taken[img], iou, -1.0)
            j = int(cand.argmax())
            if cand[j] >= thr:
                taken[img][j] = True
                tp[k] = 1
        if abs(thr - 0.5) < 1e-9:
            tp50 = int(tp.sum())
        if n_gt == 0 or len(dets) == 0:
            aps[thr] = 0.0
            continue
        ctp = np.cumsum(tp)
        recall = ctp / n_gt
        precision = ctp / (np.arange(len(dets)) + 1)
        for k in range(len(precision) - 2, -1, -1):            # precision envelope
            precision[k] = max(precision[k], precision[k + 1])
        pos = np.searchsorted(recall, recall_points, side="left")
        aps[thr] = float(np.mean([precision[p] if p < len(precision) else 0.0 for p in pos]))
    prec = tp50 / max(len(dets), 1)
    rec = tp50 / max(n_gt, 1)
    return {"mAP": float(np.mean(list(aps.values()))), "mAP50": aps.get(0.5, 0.0), "mAP75": aps.get(0.75, 0.0),
            "precision50": prec, "recall50": rec, "f1_50": 2 * prec * rec / max(prec + rec, 1e-12),
            "num_detections": len(dets), "num_ground_truth": n_gt}


# ------------------------------------------------------------------------------------------------ the model --
def build_adapted_model(config: Dict, weights_path: Optional[str], device, eval_mode: bool = True,
                        use_base_model: bool = False, dropout: Optional[float] = None) -> nn.Module:
    """Model of the training config with the adapters of ``weights_path`` loaded (``infer_sam.py:99-137``)."""
    model = default_model_builder(dict(config, engine=dict(config.get("engine") or {}, match_once=False)), device)
    if not use_base_model:
        section = dict(config["lora"])
        if dropout is not None:
            section["dropout"] = dropout
        model = apply_lora_to_model(model, LoRAConfig(**{k: section[k] for k in LORA_KEYS}))
        if weights_path is not None:
            load_lora_weights(model, weights_path)
    model.to(device)
    if (config.get("engine") or {}).get("bf16_frozen"):
        from .vit import to_training_layout
        to_training_layout(model)
    return model.eval() if eval_mode else model


def merge_lora_into_linear(model: nn.Module) -> int:
    """Replace every root-API ``LoRALinear`` by a plain ``nn.Linear`` holding ``W + (alpha/r) (A B)^T``: the exported model
    needs no adapter code.  Returns the number of merged modules."""
    from .functional import LAYOUT_ROOT, merge_weight
    todo = [(n, m) for n, m in model.named_modules() if isinstance(m, LoRALinear)]
    for name, mod in todo:
        base = mod.original_layer
        merged = merge_weight(base.weight, mod.lora.lora_A, mod.lora.lora_B, mod.lora.scaling, LAYOUT_ROOT)
        lin = nn.Linear(base.in_features, base.out_features, bias=base.bias is not None, device=base.weight.device,
                        dtype=base.weight.dtype)
        with torch.no_grad():
            lin.weight.copy_(merged.to(base.weight.dtype))
            if base.bias is not None:
                lin.bias.copy_(base.bias)
        lin.requires_grad_(False)
        parent = model
        *path, leaf = name.split(".")
        for part in path:
            parent = getattr(parent, part)
        setattr(parent, leaf, lin)
    return len(todo)


class SAM3LoRAInference:
    """SAM3 model with LoRA for inference: ``predict(image_path, text_prompts) -> {prompt index: detections}``."""

    def __init__(self, config_path: str, weights_path: Optional[str] = None, resolution: int = 1008,
                 detection_threshold: float = 0.5, device: str = "cuda", merge: bool = False):
        with open(config_path) as f:
            self.config = yaml.safe_load(f)
        if weights_path is None:
            out_dir = (self.config.get("output") or {}).get("output_dir", "outputs/sam3_lora_full")
            weights_path = os.path.join(out_dir, "best_lora_weights.pt")
            print(f"Auto-detected weights: {weights_path}")
        if not os.path.exists(weights_path):
            raise FileNotFoundError(f"LoRA weights not found: {weights_path}")
        self.weights_path, self.resolution, self.detection_threshold = weights_path, resolution, detection_threshold
        self.device = torch.device(device if torch.cuda.is_available() else "cpu")
        print(f"Initializing SAM3 + LoRA on {self.device} at {resolution}x{resolution}, threshold {detection_threshold}")
        self.model = build_adapted_model(self.config, weights_path, self.device, eval_mode=True, dropout=0.0)
        if merge:
            print(f"Merged {merge_lora_into_linear(self.model)} adapters into their frozen Linears")

    def create_datapoint(self, pil_image, text_prompts: List[str]) -> Datapoint:
        w, h = pil_image.size
        R = self.resolution
        from PIL import Image as PILImage
        resized = pil_image.resize((R, R), PILImage.BILINEAR)
        data = (torch.from_numpy(np.asarray(resized).copy()).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5
        queries = [FindQueryLoaded(query_text=t, image_id=0, object_ids_output=[], is_exhaustive=True,
                                   query_processing_order=0,
                                   inference_metadata=InferenceMetadata(coco_image_id=i, original_image_id=i,
                                                                        original_category_id=1, original_size=(w, h),
                                                                        object_id=0, frame_index=0))
                   for i, t in enumerate(text_prompts)]
        return Datapoint(find_queries=queries, images=[Image(data=data, objects=[], size=(h, w))])

    @torch.no_grad()
    def predict(self, image_path: str, text_prompts: List[str]) -> dict:
        if not os.path.exists(image_path):
            raise FileNotFoundError(f"Image not found: {image_path}")
        from PIL import Image as PILImage
        pil = PILImage.open(image_path).convert("RGB")
        return self.predict_image(pil, text_prompts)

    @torch.no_grad()
    def predict_image(self, pil, text_prompts: List[str]) -> dict:
        orig_w, orig_h = pil.size
        results = {}
        for qi, prompt in enumerate(text_prompts):           # one prompt per forward, as the reference does
            batch = collate_fn_api([self.create_datapoint(pil, [prompt])], dict_key="input")["input"]
            batch = move_to_device(batch, self.device)
            out = self.model(batch)[-1]
            scores = out["pred_logits"].float().sigmoid()[0].max(dim=-1)[0]
            keep = scores > self.detection_threshold
            n = int(keep.sum().item())
            if n == 0:
                results[qi] = {"prompt": prompt, "boxes": None, "scores": None, "masks": None, "num_detections": 0}
                continue
            cx, cy, w, h = out["pred_boxes"][0, keep].float().unbind(-1)
            boxes = torch.stack([(cx - w / 2) * orig_w, (cy - h / 2) * orig_h, (cx + w / 2) * orig_w, (cy + h / 2) * orig_h], -1)
            masks = None
            if out.get("pred_masks") is not None:
                small = out["pred_masks"][0, keep].float().sigmoid() > 0.5
                masks = (F.interpolate(small[None].float(), size=(orig_h, orig_w), mode="bilinear", align_corners=False)[0]
                         > 0.5).cpu().numpy()
            results[qi] = {"prompt": prompt, "boxes": boxes.cpu().numpy(), "scores": scores[keep].cpu().numpy(),
                           "masks": masks, "num_detections": n}
        results["_image"] = pil
        return results

    def visualize(self, results: dict, output_path: str, show_boxes: bool = True, show_masks: bool = True) -> int:
        """Overlay masks / boxes / labels on the image and save it (PIL only).  Returns the number of detections drawn."""
        from PIL import ImageDraw
        img = results["_image"].convert("RGBA")
        palette = [(255, 0, 0), (0, 0, 255), (0, 200, 0), (255, 220, 0), (0, 220, 220), (220, 0, 220)]
        total = 0
        for qi in sorted(k for k in results if k != "_image"):
            r = results[qi]
            color = palette[qi % len(palette)]
            for i in range(r["num_detections"]):
                total += 1
                if show_masks and r["masks"] is not None:
                    layer = np.zeros((*r["masks"][i].shape, 4), dtype=np.uint8)
                    layer[r["masks"][i]] = (*color, 100)
                    from PIL import Image as PILImage
                    img = PILImage.alpha_composite(img, PILImage.fromarray(layer, "RGBA"))
                if show_boxes and r["boxes"] is not None:
                    x1, y1, x2, y2 = [float(v) for v in r["boxes"][i]]
                    d = ImageDraw.Draw(img)
                    d.rectangle([max(0, x1), max(0, y1), min(img.size[0], x2), min(img.size[1], y2)], outline=color, width=2)
                    d.text((max(0, x1), max(0, y1 - 12)), f"{r['prompt']}: {r['scores'][i]:.2f}", fill=color)
        img.convert("RGB").save(output_path)
        return total


# ---------------------------------------------------------------------------------------------- validation --
class _SplitDir(COCOSegmentDataset):
    """COCO split given by its own directory (``--val_data_dir .../valid``)."""

    def __init__(self, split_dir: str, resolution: int = 1008):
        from pathlib import Path
        p = Path(split_dir)
        super().__init__(str(p.parent), p.name, resolution=resolution)


@torch.no_grad()
def validate(config_path: Optional[str], weights_path: Optional[str], val_data_dir: str, num_samples: Optional[int] = None,
             prob_threshold: float = 0.3, nms_iou: float = 0.7, merge_cracks: bool = False, merge_iou: float = 0.15,
             use_base_model: bool = False, device: Optional[str] = None, dataset=None) -> Dict[str, float]:
    """Metrics of a trained adapter file over a validation split (see the module docstring for what is computed)."""
    dev = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
    print(f"Using device: {dev}")
    if use_base_model:
        config = {"training": {"batch_size": 1}}
        if config_path:
            config = yaml.safe_load(open(config_path))
    else:
        if config_path is None or weights_path is None:
            raise ValueError("config_path and weights_path are required when use_base_model=False")
        config = yaml.safe_load(open(config_path))
    model = build_adapted_model(config, weights_path, dev, eval_mode=True, use_base_model=use_base_model)
    stats = count_parameters(model)
    print(f"Total params: {stats['total_parameters']:,}; adapter params: {stats['trainable_parameters']:,}")
    ds = dataset if dataset is not None else _SplitDir(val_data_dir)
    n = len(ds) if num_samples is None else min(num_samples, len(ds))
    bs = int(config["training"].get("batch_size", 1))
    detections, ground_truth = [], {}
    for start in range(0, n, bs):
        samples = [ds[i] for i in range(start, min(start + bs, n))]
        batch = move_to_device(collate_fn_api(samples, dict_key="input", with_seg_masks=True)["input"], dev)
        out = model(batch)[-1]
        for k, sample in enumerate(samples):
            img_id = start + k
            mp, sc, bx = apply_sam3_nms(out["pred_logits"][k], out["pred_masks"][k], out["pred_boxes"][k], prob_threshold, nms_iou)
            binm = mp > 0.5
            if merge_cracks:
                binm, sc, bx = merge_overlapping_masks(binm, sc, bx, merge_iou)
            gts = [o.segment for o in sample.images[0].objects if o.segment is not None]
            size = gts[0].shape if gts else binm.shape[-2:]
            if len(binm) and tuple(binm.shape[-2:]) != tuple(size):      # predictions at 288^2 -> target resolution
                binm = F.interpolate(binm[None].float(), size=tuple(size), mode="bilinear", align_corners=False)[0] > 0.5
            ground_truth[img_id] = [g.cpu().numpy() for g in gts]
            for m, s in zip(binm, sc):
                detections.append((img_id, float(s), m.cpu().numpy()))
    metrics = mask_average_precision(detections, ground_truth)
    metrics["images"] = n
    print("\nValidation (mask AP, COCO definition; single category = the query text):")
    for k in ("mAP", "mAP50", "mAP75", "precision50", "recall50", "f1_50", "num_detections", "num_ground_truth", "images"):
        v = metrics[k]
        print(f"  {k:18s} {v:.4f}" if isinstance(v, float) else f"  {k:18s} {v}")
    return metrics
