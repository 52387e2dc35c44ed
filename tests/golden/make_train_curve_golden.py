#!/usr/bin/env python3
"""
Loss curve of the REFERENCE over a few optimizer steps (SURVEY section 8c "Model step": A/B after AdamW; north star:
"loss curve matching CPU reference within 1e-3"): the reference ViT class at the tiny configuration of vit_tiny.npz,
the reference root LoRA injector (r=4, alpha=8, fc1/fc2, dropout 0), torch AdamW exactly as the native CLI builds it
(``train_sam3_lora_native.py:736-740``: params with requires_grad, default betas/eps) at lr 2e-3 / wd 0.01, 12 steps on
one fixed seeded batch with loss = mean((feat - target)^2), train mode (activation checkpointing on), CPU fp32.
Build container only.

    python tests/golden/make_train_curve_golden.py      -> train_curve.npz  (losses[12], final A/B of every adapter)
"""
import contextlib
import io
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
from make_vit_golden import TINY

STEPS, LR, WD = 12, 2e-3, 0.01


def main():
    sam3_manifest._install_stubs()
    sys.modules["timm.layers"].trunc_normal_ = torch.nn.init.trunc_normal_
    from sam3.model.vitdet import ViT
    import lora_layers as ref_root
    assert ref_root.__file__.startswith(REF)
    g0 = np.load(os.path.join(HERE, "vit_tiny.npz"))
    vit = ViT(**TINY)
    sd = {}
    for k in g0.files:
        if k.startswith("sd/"):
            name = k[3:]
            if name.endswith(".re"):
                sd[name[:-3]] = torch.complex(torch.from_numpy(g0[k]), torch.from_numpy(g0["sd/" + name[:-3] + ".im"]))
            elif not name.endswith(".im"):
                sd[name] = torch.from_numpy(g0[k])
    vit.load_state_dict(sd, strict=True)
    with contextlib.redirect_stdout(io.StringIO()):
        ref_root.apply_lora_to_model(vit, ref_root.LoRAConfig(rank=4, alpha=8, dropout=0.0, target_modules=["fc1", "fc2"]))
    with torch.no_grad():
        for n, m in vit.named_modules():
            if isinstance(m, ref_root.LoRALayer):
                m.lora_A.copy_(torch.from_numpy(g0[f"lora/{n}.lora_A"]))
                m.lora_B.copy_(torch.from_numpy(g0[f"lora/{n}.lora_B"]))
    img = torch.from_numpy(g0["img"])
    gen = torch.Generator().manual_seed(77)
    target = torch.randn(2, 64, 8, 8, generator=gen) * 0.5
    opt = torch.optim.AdamW([p for p in vit.parameters() if p.requires_grad], lr=LR, weight_decay=WD)
    vit.train()
    losses = []
    for _ in range(STEPS):
        loss = ((vit(img)[0] - target) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out = {"losses": np.array(losses, np.float64), "target": target.numpy(), "lr": np.float64(LR), "wd": np.float64(WD)}
    for n, m in vit.named_modules():
        if isinstance(m, ref_root.LoRALayer):
            out[f"A/{n}"] = m.lora_A.detach().numpy().copy()
            out[f"B/{n}"] = m.lora_B.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "train_curve.npz"), **out)
    print("losses:", " ".join(f"{l:.6f}" for l in losses))


if __name__ == "__main__":
    main()
