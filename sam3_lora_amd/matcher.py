"""
Hungarian matching of DETR queries to ground-truth boxes -- the parity surface right after the model
forward in the training step (reference ``sam3/train/matcher.py``:
``BinaryHungarianMatcherV2`` :431-668, ``_do_matching`` :15-29, box utilities ``sam3/model/box_ops.py``
:11-143).  Indices must be bit-exact, so the assignment itself stays on the host with the same solver
(``scipy.optimize.linear_sum_assignment``); the cost matrix is built on the device in fp32 in ONE batched
expression and crosses to the host once per call.

    cost = w_bbox * L1(cxcywh) + w_class * focal_class_cost + w_giou * (-GIoU)

The native CLI constructs it as ``BinaryHungarianMatcherV2(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0,
focal=True)`` (``train_sam3_lora_native.py:743-745``).
"""
from __future__ import annotations
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

__all__ = ["BinaryHungarianMatcherV2", "box_cxcywh_to_xyxy", "box_iou", "generalized_box_iou"]

INVALID_COST = 1e9      # cost given to masked-out predictions / targets
VALID_THRESH = 1e8      # assignments at or above this are dropped after the solve


def box_cxcywh_to_xyxy(x: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_iou(a: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pairwise IoU and union of xyxy boxes a[..., N, 4], b[..., M, 4] -> [..., N, M]."""
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    lt = torch.max(a[..., :, None, :2], b[..., None, :, :2])
    rb = torch.min(a[..., :, None, 2:], b[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[..., None] + area_b[..., None, :] - inter
    return inter / union, union


def generalized_box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    iou, union = box_iou(a, b)
    lt = torch.min(a[..., :, None, :2], b[..., None, :, :2])
    rb = torch.max(a[..., :, None, 2:], b[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


def _solve(cost: np.ndarray, repeats: int, want_tgt: bool, filter_invalid: bool):
    """One image: rows = queries, cols = its targets (tiled ``repeats`` times for one-to-many)."""
    if repeats > 1:
        cost = np.tile(cost, (1, repeats))
    if not np.isfinite(cost).all():
        # the reference lets scipy raise here (a diverged run ends); in the fp8 frozen-W mode a non-finite forward is an event the
        # trainer survives -- the loss and every gradient of the step are non-finite too and the optimizer step is skipped
        # (trainer.NonFiniteStepGuard) -- so the assignment only has to exist
        from . import fp8
        if not fp8.fp8_enabled():
            raise ValueError("matrix contains invalid numeric entries")
        cost = np.nan_to_num(cost, nan=1e9, posinf=1e9, neginf=-1e9)
    rows, cols = linear_sum_assignment(cost)
    if filter_invalid:
        keep = cost[rows, cols] < VALID_THRESH
        rows, cols = rows[keep].astype(np.int64), cols[keep].astype(np.int64)
    if want_tgt:
        return rows, cols
    return rows[np.argsort(cols)]       # query index of target 0, 1, 2, ...


class BinaryHungarianMatcherV2(nn.Module):
    """Returns ``(batch_idx, src_idx, tgt_idx)``; ``tgt_idx`` is None when every image has fewer targets
    than queries and nothing is masked (then matches are listed in target order per image)."""

    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, focal: bool = False,
                 alpha: float = 0.25, gamma: float = 2.0, stable: bool = False,
                 remove_samples_with_0_gt: bool = True):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.norm = nn.Sigmoid()
        self.focal = focal
        if focal:
            self.alpha, self.gamma, self.stable = alpha, gamma, stable
        self.remove_samples_with_0_gt = remove_samples_with_0_gt

    def cost_matrix(self, score: torch.Tensor, boxes: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        """score [B, Q] logits, boxes [B, Q, 4] cxcywh, tgt [B, T, 4] cxcywh (padded) -> [B, Q, T]."""
        c_l1 = torch.cdist(boxes, tgt, p=1)
        c_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt))
        prob = self.norm(score)
        if not self.focal:
            c_cls = -prob.unsqueeze(-1).expand_as(c_l1)
        elif self.stable:
            p = prob.unsqueeze(-1).expand_as(c_l1) * ((-c_giou + 1) / 2)
            c_cls = -self.alpha * (1 - p) ** self.gamma * torch.log(p) + (1 - self.alpha) * p ** self.gamma * torch.log(1 - p)
        else:
            c_cls = (-self.alpha * (1 - prob) ** self.gamma * torch.nn.functional.logsigmoid(score)
                     + (1 - self.alpha) * prob ** self.gamma * torch.nn.functional.logsigmoid(-score))
            c_cls = c_cls.unsqueeze(-1).expand_as(c_l1)
        return self.cost_bbox * c_l1 + self.cost_class * c_cls + self.cost_giou * c_giou

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], batched_targets: Dict[str, torch.Tensor], repeats: int = 1,
                repeat_batch: int = 1, out_is_valid: Optional[torch.Tensor] = None,
                target_is_valid_padded: Optional[torch.Tensor] = None):
        return self.collect(self.launch([outputs], batched_targets, repeats=repeats, repeat_batch=repeat_batch,
                                        out_is_valid=out_is_valid, target_is_valid_padded=target_is_valid_padded))[0]

    @torch.no_grad()
    def launch(self, outputs_list: Sequence[Dict[str, torch.Tensor]], batched_targets: Dict[str, torch.Tensor],
               repeats: int = 1, repeat_batch: int = 1, out_is_valid: Optional[torch.Tensor] = None,
               target_is_valid_padded: Optional[torch.Tensor] = None) -> Dict:
        """First half of matching ``len(outputs_list)`` outputs (a decoder's final + auxiliary ones) against the same
        targets: the cost of ALL of them as one batched fp32 expression on the device and ONE device->host copy,
        asynchronous on a GPU (pinned buffer + event).  Nothing here waits for the device when the targets carry their
        box counts on the host (``num_boxes_host``, sam3_data.collate_fn_api), so the caller can keep queueing device
        work -- the mask head -- and ``collect`` later."""
        L = len(outputs_list)
        num_queries = outputs_list[0]["pred_logits"].shape[1]
        # the cost is an fp32 expression whatever the model's output dtype (bf16 heads in the MI355X layout);
        # numpy has no bfloat16 and the assignment must not depend on the activation dtype
        score = torch.cat([o["pred_logits"].squeeze(-1).float() for o in outputs_list], 0)
        boxes = torch.cat([o["pred_boxes"].float() for o in outputs_list], 0)
        device = score.device
        host_counts = batched_targets.get("num_boxes_host")
        num_boxes = (torch.as_tensor(list(host_counts), dtype=torch.long) if host_counts is not None
                     else batched_targets["num_boxes"].cpu())
        tgt = batched_targets["boxes_padded"].float()
        keep = None
        if self.remove_samples_with_0_gt:
            keep = num_boxes > 0
            if bool(keep.all()):
                keep_dev = None
            else:
                keep_dev = keep.to(device)
                tgt = tgt[keep_dev]
            num_boxes = num_boxes[keep]
            if target_is_valid_padded is not None and keep_dev is not None:
                target_is_valid_padded = target_is_valid_padded[keep_dev]
        reps = repeat_batch * L         # final + auxiliary outputs concatenated along the batch
        if reps > 1:
            num_boxes = num_boxes.repeat(reps)
            tgt = tgt.repeat(reps, 1, 1)
            if target_is_valid_padded is not None:
                target_is_valid_padded = target_is_valid_padded.repeat(reps, 1)
        if self.remove_samples_with_0_gt:
            if reps > 1:
                keep = keep.repeat(reps)
            if keep_dev is not None:
                kd = keep.to(device)
                score, boxes = score[kd], boxes[kd]
                if out_is_valid is not None:
                    out_is_valid = out_is_valid[kd]
        assert boxes.shape[0] == tgt.shape[0] == num_boxes.shape[0]

        C = self.cost_matrix(score, boxes, tgt)
        filtering = out_is_valid is not None or target_is_valid_padded is not None
        if out_is_valid is not None:
            C = torch.where(out_is_valid[:, :, None], C, INVALID_COST)
        if target_is_valid_padded is not None:
            C = torch.where(target_is_valid_padded[:, None, :], C, INVALID_COST)
        event = None
        if C.is_cuda:                                                # the one device->host crossing
            C_host = torch.empty(C.shape, dtype=C.dtype, pin_memory=True)
            C_host.copy_(C, non_blocking=True)
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(device))
        else:
            C_host = C
        return dict(cost=C_host, event=event, num_boxes=num_boxes, keep=keep, L=L, repeats=repeats,
                    filtering=filtering, num_queries=num_queries, device=device)

    @torch.no_grad()
    def collect(self, handle: Dict) -> List[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]]:
        """Second half: wait for the cost copy, solve every image on the host, return one index triple per output."""
        if handle["event"] is not None:
            handle["event"].synchronize()
        C = handle["cost"].numpy()
        L, repeats, filtering, device = handle["L"], handle["repeats"], handle["filtering"], handle["device"]
        num_boxes, keep = handle["num_boxes"], handle["keep"]
        counts = num_boxes.tolist()
        per_image = [C[i, :, :n] for i, n in enumerate(counts)]
        want_tgt = filtering or bool(torch.any(handle["num_queries"] < num_boxes * max(repeats, 1)).item())
        solved = [_solve(c, repeats, want_tgt, filtering) for c in per_image]
        image_ids_all = keep.nonzero().squeeze(1).tolist() if self.remove_samples_with_0_gt else list(range(len(solved)))
        n_img = len(solved) // L                 # rows per output (images kept x repeat_batch)
        total_rows = (len(keep) // L) if keep is not None else n_img
        results, upload = [], []
        for l in range(L):
            rows = solved[l * n_img:(l + 1) * n_img]
            cnt = counts[l * n_img:(l + 1) * n_img]
            ids = [i - l * total_rows for i in image_ids_all[l * n_img:(l + 1) * n_img]]
            empty = np.zeros(0, dtype=np.int64)
            if not rows:
                src_lists, tgt_idx = [], (empty if want_tgt else None)
            elif want_tgt:
                src_lists = [s for s, _ in rows]
                offsets = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
                tgt_idx = np.concatenate([t + o for (_, t), o in zip(rows, offsets)]).astype(np.int64)
            else:
                src_lists, tgt_idx = rows, None
            batch_idx = np.asarray([ids[i] for i, s in enumerate(src_lists) for _ in range(len(s))], dtype=np.int64)
            src_idx = np.concatenate(src_lists).astype(np.int64) if src_lists else empty
            upload.append((batch_idx, src_idx, tgt_idx))
        # ONE host -> device transfer for every index list of every output (pinned, asynchronous on a GPU: a pageable
        # copy would block the host until the stream -- with the mask head queued on it -- has drained)
        flat = [a for trip in upload for a in trip if a is not None]
        sizes = [len(a) for a in flat]
        packed = torch.from_numpy(np.concatenate(flat) if flat else np.zeros(0, dtype=np.int64))
        if device.type == "cuda":
            packed = packed.pin_memory().to(device, non_blocking=True)
        parts = iter(packed.split(sizes)) if flat else iter(())
        for b_, s_, t_ in upload:
            results.append((next(parts), next(parts), None if t_ is None else next(parts)))
        return results
