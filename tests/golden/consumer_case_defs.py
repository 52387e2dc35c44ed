"""Inputs shared by make_consumer_golden.py (runs the REFERENCE's validate_sam3_lora.py functions) and tests/test_inference.py
(runs sam3_lora_amd.inference): seeded detections with clustered, overlapping masks.  Data and plain numbers only."""
import torch

# (name, number of detections, mask side, clusters, seed, prob_threshold, nms_iou_threshold, max_detections, merge_iou_threshold)
CASES = [
    ("clustered", 40, 32, 6, 1, 0.3, 0.7, 100, 0.15),
    ("tight_nms", 40, 32, 6, 2, 0.3, 0.3, 100, 0.15),
    ("topk", 60, 24, 10, 3, 0.1, 0.9, 5, 0.05),
    ("all_below_threshold", 12, 16, 3, 4, 0.999, 0.7, 100, 0.15),
    ("single", 1, 16, 1, 5, 0.0, 0.7, 100, 0.15),
    ("aggressive_merge", 30, 32, 4, 6, 0.2, 0.7, 100, 0.01),
]


def make_case(n, side, clusters, seed):
    """(pred_logits [n, 1], pred_masks [n, side, side] mask LOGITS, pred_boxes [n, 4]): rectangles jittered around `clusters`
    centres (so that many pairs overlap at IoU 0.1 .. 0.9), logits = +-4 inside / outside plus noise; distinct scores."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(clusters, 2, generator=g) * 0.6 + 0.2
    which = torch.randint(0, clusters, (n,), generator=g)
    c = centres[which] + (torch.rand(n, 2, generator=g) - 0.5) * 0.12
    wh = torch.rand(n, 2, generator=g) * 0.25 + 0.12
    boxes = torch.cat([c, wh], -1)
    ys = (torch.arange(side) + 0.5) / side
    inside = lambda lo, hi: (ys[None] >= lo[:, None]) & (ys[None] < hi[:, None])
    m = inside(c[:, 1] - wh[:, 1] / 2, c[:, 1] + wh[:, 1] / 2)[:, :, None] & inside(c[:, 0] - wh[:, 0] / 2, c[:, 0] + wh[:, 0] / 2)[:, None, :]
    masks = torch.where(m, 4.0, -4.0) + torch.randn(n, side, side, generator=g) * 1.5
    logits = (torch.randperm(n, generator=g).float() / n * 8 - 4).unsqueeze(-1)         # distinct scores: no ties in the sort
    return logits, masks, boxes
