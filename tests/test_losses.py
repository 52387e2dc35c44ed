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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("N,h,w,H,W", [(3, 16, 16, 40, 40), (5, 288, 288, 1008, 1008), (2, 20, 24, 70, 50), (1, 8, 8, 8, 8),
                                       (4, 32, 32, 112, 112)])
def test_mask_loss_kernels_match_the_pytorch_formulation(N, h, w, H, W, dtype):
    """sam3_mask_loss_fwd / _bwd (upsample + focal + dice in one pass, gather backward) against F.interpolate + the
    formulas of losses.py in fp32: both loss values to 2e-5 relative, the logit gradient to 2e-4 of its max (fp32
    logits; bf16 logits: the same inputs rounded, the gradient rounded to bf16 -> 4e-3), bit-identical run to run."""
    dev = "cuda:0"
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(N * 1000 + h)
    src = (torch.randn(N, h, w, device=dev, generator=g) * 3).to(td)
    tgt = torch.rand(N, H, W, device=dev, generator=g) > 0.6
    tgt[0, : H // 2] = True                              # one large object: dice terms of very different sizes
    nb = 2.0
    for alpha, gamma in ((0.25, 2.0), (-1.0, 1.5)):
        a = src.clone().requires_grad_(True)
        lm, ld = LS.mask_losses_fused(a, tgt, nb, alpha, gamma)
        (200.0 * lm + 10.0 * ld).backward()
        b = src.float().clone().requires_grad_(True)
        up = torch.nn.functional.interpolate(b[:, None], size=(H, W), mode="bilinear", align_corners=False)[:, 0].flatten(1)
        t = tgt.flatten(1).float()
        lm_ref = LS.sigmoid_focal_loss(up, t, nb, alpha=alpha, gamma=gamma)
        ld_ref = LS.dice_loss(up, t, nb)
        (200.0 * lm_ref + 10.0 * ld_ref).backward()
        assert abs(lm.item() - lm_ref.item()) <= 2e-5 * abs(lm_ref.item()) + 1e-7, (lm.item(), lm_ref.item())
        assert abs(ld.item() - ld_ref.item()) <= 2e-5 * abs(ld_ref.item()) + 1e-7, (ld.item(), ld_ref.item())
        err = (a.grad.float() - b.grad).abs().max().item() / b.grad.abs().max().item()
        assert err <= (2e-4 if dtype == "f32" else 4e-3), err
        assert a.grad.dtype == td
        a2 = src.clone().requires_grad_(True)
        lm2, ld2 = LS.mask_losses_fused(a2, tgt, nb, alpha, gamma)
        (200.0 * lm2 + 10.0 * ld2).backward()
        assert lm2.item() == lm.item() and ld2.item() == ld.item() and torch.equal(a2.grad, a.grad)


@pytest.mark.gpu
def test_masks_loss_kernel_and_pytorch_paths_agree_in_the_loss_stack(golden_dir):
    """The Masks loss inside the wrapper: kernel path (default on the GPU) against use_kernel = False."""
    cfg = CLI_LOSS_CFG
    res = []
    for use_kernel in (True, False):
        matcher = BinaryHungarianMatcherV2(**cfg["matcher"])
        masks = LS.Masks(**cfg["masks"])
        masks.use_kernel = use_kernel
        leaves, targets = make_raw()
        leaves = {k: v.to("cuda:0").requires_grad_(True) for k, v in leaves.items()}
        targets = {k: v.to("cuda:0") for k, v in targets.items()}
        out = assemble(leaves)
        idx = matcher(out, targets)
        d = masks(outputs=out, targets=targets, indices=idx, num_boxes=3.0)
        d[LS.CORE_LOSS_KEY].backward()
        res.append((d, leaves["main/masks"].grad.clone()))
    for k in ("loss_mask", "loss_dice", LS.CORE_LOSS_KEY):
        assert abs(float(res[0][0][k]) - float(res[1][0][k])) <= 2e-5 * abs(float(res[1][0][k]))
    assert (res[0][1] - res[1][1]).abs().max() <= 2e-4 * res[1][1].abs().max()


@pytest.mark.gpu
def test_box_pair_kernels_match_the_elementwise_formulation():
    """sam3_box_pair_fwd / _bwd behind diag_box_iou / diag_generalized_box_iou on the GPU against the operator-by-operator
    expressions (the CPU path of the same functions) in fp64: values and the gradient with respect to the predictions,
    for overlapping, nested, disjoint, touching and identical (tie) boxes."""
    from sam3_lora_amd.losses import diag_box_iou, diag_generalized_box_iou
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(64, 2, generator=g) * 0.6
    wh = torch.rand(64, 2, generator=g) * 0.35 + 0.02
    a = torch.cat([xy, xy + wh], 1)
    xy2 = torch.rand(64, 2, generator=g) * 0.6
    wh2 = torch.rand(64, 2, generator=g) * 0.35 + 0.02
    b = torch.cat([xy2, xy2 + wh2], 1)
    b[:4] = a[:4]                                                     # identical boxes: every min / max is a tie
    b[4:8] = torch.tensor([0.30, 0.30, 0.40, 0.40]); a[4:8] = torch.tensor([0.10, 0.10, 0.90, 0.80])    # nested
    a[8:12] = torch.tensor([0.05, 0.05, 0.15, 0.15]); b[8:12] = torch.tensor([0.50, 0.60, 0.90, 0.95])  # disjoint
    a[12:14] = torch.tensor([0.10, 0.10, 0.30, 0.30]); b[12:14] = torch.tensor([0.30, 0.10, 0.50, 0.30])  # touching edge
    w = torch.randn(64, 2, generator=g)
    ad = a.double().requires_grad_(True)
    ref = (diag_box_iou(ad, b.double()) * w[:, 0].double() + diag_generalized_box_iou(ad, b.double()) * w[:, 1].double())
    ref.sum().backward()
    ac = a.cuda().requires_grad_(True)
    got = diag_box_iou(ac, b.cuda()) * w[:, 0].cuda() + diag_generalized_box_iou(ac, b.cuda()) * w[:, 1].cuda()
    assert "BoxPair" in type(diag_box_iou(ac, b.cuda()).grad_fn.next_functions[0][0]).__name__      # the kernel path ran
    got.sum().backward()
    assert torch.allclose(got.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(ac.grad.cpu().double(), ad.grad, rtol=1e-4, atol=1e-5)
    # empty match lists stay on the operator path
    e = torch.zeros(0, 4, device="cuda")
    assert diag_generalized_box_iou(e, e).shape == (0,)
