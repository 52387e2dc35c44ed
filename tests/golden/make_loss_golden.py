#!/usr/bin/env python3
"""Golden values for the training losses: the REFERENCE Boxes / IABCEMdetr / Masks /
Sam3LossWrapper.compute_loss / matchers (imported from /root/reference, CPU fp32, focal loss with triton=False --
SURVEY F7) on the synthetic outputs of loss_case_defs.py, configured exactly as train_sam3_lora_native.py:743-793.
Stores every returned scalar and the gradient of core_loss wrt each prediction leaf.  Build container only."""
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)
import numpy as np
import torch
import sam3_manifest
from loss_case_defs import CLI_LOSS_CFG, assemble, make_raw


def main():
    sam3_manifest._install_stubs()
    tm = sys.modules.get("torchmetrics.functional") or __import__("torchmetrics.functional", fromlist=["x"])
    tm.f1_score = lambda *a, **k: torch.tensor(0.0)
    import torchmetrics
    torchmetrics.functional = tm
    from sam3.train.loss import loss_fns as LF
    from sam3.train.loss.sam3_loss import Sam3LossWrapper
    from sam3.train.matcher import BinaryHungarianMatcherV2, BinaryOneToManyMatcher
    LF.sigmoid_focal_loss = functools.partial(LF.sigmoid_focal_loss, triton=False)

    cfg = CLI_LOSS_CFG
    matcher = BinaryHungarianMatcherV2(**cfg["matcher"])
    wrapper = Sam3LossWrapper(loss_fns_find=[LF.Boxes(**cfg["boxes"]), LF.IABCEMdetr(**cfg["ce"]), LF.Masks(**cfg["masks"])],
                              matcher=matcher, o2m_matcher=BinaryOneToManyMatcher(**cfg["o2m"]), **cfg["wrapper"])
    leaves, targets = make_raw()
    for v in leaves.values():
        v.requires_grad_(True)
    out = assemble(leaves)
    out["indices"] = matcher(out, targets)
    for a in out["aux_outputs"]:
        a["indices"] = matcher(a, targets)
    losses = wrapper.compute_loss(out, targets)
    losses["core_loss"].backward()
    res = {}
    for k, v in losses.items():
        res[f"loss/{k}"] = np.float64(float(v))
    for k, v in leaves.items():
        res[f"grad/{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
    bi, si, ti = out["indices"]
    res["main_indices"] = np.stack([bi.numpy(), si.numpy()])
    o2m = BinaryOneToManyMatcher(**cfg["o2m"])({k[:-4]: v for k, v in out.items() if k.endswith("_o2m")}, targets)
    res["o2m_indices"] = np.stack([t.numpy() for t in o2m])
    np.savez_compressed(os.path.join(HERE, "loss_cases.npz"), **res)
    for k in sorted(losses):
        print(f"{k:28s} {float(losses[k]):.6f}")


if __name__ == "__main__":
    main()
