"""Losses + one-to-many matcher + wrapper (sam3_lora_amd/losses.py) against the reference's values on the same
synthetic outputs/targets (tests/golden/loss_cases.npz): every named scalar, core_loss, and the gradient of
core_loss wrt every prediction tensor; the o2m match indices bit-exact."""
import os

import numpy as np
import pytest
import torch

from loss_case_defs import CLI_LOSS_CFG, assemble, make_raw
from sam3_lora_amd import losses as LS
from sam3_lora_amd.matcher import BinaryHungarianMatcherV2


def _run(device):
    cfg = CLI_LOSS_CFG
    matcher = BinaryHungarianMatcherV2(**cfg["matcher"])
    o2m = LS.BinaryOneToManyMatcher(**cfg["o2m"])
    wrapper = LS.Sam3LossWrapper([LS.Boxes(**cfg["boxes"]), LS.IABCEMdetr(**cfg["ce"]), LS.Masks(**cfg["masks"])],
                                 matcher=matcher, o2m_matcher=o2m, **cfg["wrapper"])
    leaves, targets = make_raw()
    leaves = {k: v.to(device).requires_grad_(True) for k, v in leaves.items()}
    targets = {k: v.to(device) for k, v in targets.items()}
    out = assemble(leaves)
    out["indices"] = matcher(out, targets)
    for a in out["aux_outputs"]:
        a["indices"] = matcher(a, targets)
    losses = wrapper.compute_loss(out, targets)
    losses[LS.CORE_LOSS_KEY].backward()
    o2m_idx = o2m({k[:-4]: v for k, v in out.items() if k.endswith("_o2m")}, targets)
    return losses, leaves, out["indices"], o2m_idx


def _check(golden_dir, device, rtol):
    g = np.load(os.path.join(golden_dir, "loss_cases.npz"))
    losses, leaves, idx, o2m_idx = _run(device)
    assert np.array_equal(np.stack([idx[0].cpu().numpy(), idx[1].cpu().numpy()]), g["main_indices"])
    assert np.array_equal(np.stack([t.cpu().numpy() for t in o2m_idx]), g["o2m_indices"])
    want_keys = {k[5:] for k in g.files if k.startswith("loss/")}
    assert set(losses) == want_keys
    for k in sorted(want_keys):
        if k.startswith("ce_f1"):
            continue          # logging-only metric; the golden run stubs torchmetrics
        got, ref = float(torch.as_tensor(losses[k]).detach()), float(g[f"loss/{k}"])
        assert abs(got - ref) <= rtol * max(1.0, abs(ref)), (k, got, ref)
    for k, v in leaves.items():
        ref = g[f"grad/{k}"]
        got = (v.grad if v.grad is not None else torch.zeros_like(v)).cpu().numpy()
        assert np.abs(got - ref).max() <= rtol * max(1e-3, np.abs(ref).max()), k


def test_losses_match_reference_cpu(golden_dir):
    _check(golden_dir, "cpu", 2e-5)


@pytest.mark.gpu
def test_losses_match_reference_gpu(golden_dir):
    _check(golden_dir, "cuda:0", 1e-4)


def test_focal_and_dice_edge_cases():
    x = torch.zeros(2, 8)
    t = torch.zeros(2, 8)
    assert torch.isfinite(LS.sigmoid_focal_loss(x, t, 1.0)) and torch.isfinite(LS.dice_loss(x, t, 1.0))
    # empty match set: Masks returns finite zeros-like values
    m = LS.Masks(weight_dict={"loss_mask": 1.0, "loss_dice": 1.0})
    e = torch.empty(0, dtype=torch.long)
    out = m(outputs={"pred_masks": torch.randn(1, 4, 8, 8)},
            targets={"masks": torch.zeros(0, 16, 16, dtype=torch.bool), "is_valid_mask": torch.zeros(0, dtype=torch.bool)},
            indices=(e, e, None), num_boxes=1.0)
    assert float(out["loss_mask"]) == 0.0 and float(out["loss_dice"]) == 0.0
