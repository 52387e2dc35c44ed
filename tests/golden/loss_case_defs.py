"""Synthetic model outputs / targets shared by make_loss_golden.py (reference) and tests/test_losses.py (ours)."""
import torch


def cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def make_raw(seed=7, B=3, Q=12, nb=(2, 0, 3), hw=16, HW=40, n_aux=2):
    """Leaf tensors (to be given requires_grad) + targets."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    u = lambda *s: torch.rand(*s, generator=g)
    def boxes(*lead):
        return torch.cat([u(*lead, 2) * 0.6 + 0.2, u(*lead, 2) * 0.3 + 0.05], -1)
    leaves = {}
    for tag in ["main"] + [f"aux{i}" for i in range(n_aux)]:
        for tw in ("", "_o2m"):
            leaves[f"{tag}/logits{tw}"] = r(B, Q, 1)
            leaves[f"{tag}/boxes{tw}"] = boxes(B, Q)
        leaves[f"{tag}/presence"] = r(B, 1)
    leaves["main/masks"] = r(B, Q, hw, hw)
    leaves["main/masks_o2m"] = r(B, Q, hw, hw)
    N = sum(nb)
    T = max(max(nb), 1)
    tb = boxes(N)
    padded = torch.zeros(B, T, 4)
    ids = torch.full((B, T), -1, dtype=torch.long)
    o = 0
    for b, n in enumerate(nb):
        padded[b, :n] = tb[o:o + n]
        ids[b, :n] = torch.arange(n)
        o += n
    masks = u(N, HW, HW) > 0.6
    targets = dict(boxes=tb, boxes_xyxy=cxcywh_to_xyxy(tb), boxes_padded=padded, num_boxes=torch.tensor(nb),
                   object_ids_padded=ids, is_exhaustive=torch.tensor([True, True, False][:B]),
                   masks=masks, is_valid_mask=torch.tensor([True] * (N - 1) + [False]))
    return leaves, targets


def assemble(leaves, n_aux=2):
    """The nested output dict the model would produce (without 'indices')."""
    def one(tag, with_masks):
        d = {"pred_logits": leaves[f"{tag}/logits"], "pred_boxes": leaves[f"{tag}/boxes"],
             "pred_boxes_xyxy": cxcywh_to_xyxy(leaves[f"{tag}/boxes"]), "presence_logit_dec": leaves[f"{tag}/presence"],
             "pred_logits_o2m": leaves[f"{tag}/logits_o2m"], "pred_boxes_o2m": leaves[f"{tag}/boxes_o2m"],
             "pred_boxes_xyxy_o2m": cxcywh_to_xyxy(leaves[f"{tag}/boxes_o2m"])}
        if with_masks:
            d["pred_masks"] = leaves[f"{tag}/masks"]
            d["pred_masks_o2m"] = leaves[f"{tag}/masks_o2m"]
        return d
    out = one("main", True)
    out["aux_outputs"] = [one(f"aux{i}", False) for i in range(n_aux)]
    return out


CLI_LOSS_CFG = dict(
    boxes=dict(weight_dict={"loss_bbox": 5.0, "loss_giou": 2.0}),
    ce=dict(pos_weight=10.0, weight_dict={"loss_ce": 20.0, "presence_loss": 20.0}, pos_focal=False, alpha=0.25, gamma=2,
            use_presence=True, pad_n_queries=200),
    masks=dict(weight_dict={"loss_mask": 200.0, "loss_dice": 10.0}, focal_alpha=0.25, focal_gamma=2.0, compute_aux=False),
    matcher=dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True),
    o2m=dict(alpha=0.3, threshold=0.4, topk=4),
    wrapper=dict(o2m_weight=2.0, use_o2m_matcher_on_o2m_aux=False, normalization="local"),
)
