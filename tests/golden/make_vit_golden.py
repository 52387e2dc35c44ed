#!/usr/bin/env python3
"""
Golden vectors for the ViT-Det trunk that hosts the adapters, produced by RUNNING THE
REFERENCE classes (``sam3/model/vitdet.py`` imported from /root/reference, CPU fp32) at a tiny
configuration built from the reference's own parametric constructor -- same structure as SAM3's trunk
(patch 14, tiled abs-pos with cls token, ln_pre, windowed + global blocks, interpolated 2-D RoPE,
mlp_ratio 4.625) at embed 64 / depth 4 / 8x8 tokens.  Build container only.

    python tests/golden/make_vit_golden.py

Writes ``vit_tiny.npz``:
    sd/<name>            reference state dict (weights + RoPE buffers), fp32 / complex64 split re+im
    img, gout            seeded input batch and upstream gradient
    feat                 ViT(img)[0] in eval mode (no DropPath)
    With the reference ROOT LoRA (r=4, alpha=8, fc1/fc2) injected by the reference injector and B set to
    seeded non-zero values:  lora/<name> (A, B), feat_lora, gimg, gA/<name>, gB/<name> for loss=(feat*gout).sum()
    run in train mode with drop_path_rate=0 (activation checkpointing ON, as in the reference).
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

import sam3_manifest  # harness-side stand-ins for timm / torchvision / iopath (not reference code)

TINY = dict(img_size=112, pretrain_img_size=56, patch_size=14, embed_dim=64, depth=4, num_heads=2,
            mlp_ratio=4.625, norm_layer="LayerNorm", drop_path_rate=0.0, qkv_bias=True, use_abs_pos=True,
            tile_abs_pos=True, global_att_blocks=(1, 3), rel_pos_blocks=(), use_rope=True, use_interp_rope=True,
            window_size=4, pretrain_use_cls_token=True, retain_cls_token=False, ln_pre=True, ln_post=False,
            return_interm_layers=False, bias_patch_embed=False, compile_mode=None)


def main():
    sam3_manifest._install_stubs()
    # real trunc_normal_ so the tiny model has non-degenerate weights
    sys.modules["timm.layers"].trunc_normal_ = torch.nn.init.trunc_normal_
    from sam3.model.vitdet import ViT
    import lora_layers as ref_root
    assert ref_root.__file__.startswith(REF)

    torch.manual_seed(0)
    vit = ViT(**TINY)
    # make LayerNorms / biases non-trivial (the reference initialises them to 1 / 0)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in vit.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    out = {}
    for k, v in vit.state_dict().items():
        if v.is_complex():
            out[f"sd/{k}.re"] = v.real.numpy().copy()
            out[f"sd/{k}.im"] = v.imag.numpy().copy()
        else:
            out[f"sd/{k}"] = v.numpy().copy()
    img = torch.randn(2, 3, 112, 112, generator=g)
    gout = torch.randn(2, 64, 8, 8, generator=g)
    vit.eval()
    with torch.no_grad():
        out["feat"] = vit(img)[0].numpy().copy()
    out["img"], out["gout"] = img.numpy(), gout.numpy()

    with contextlib.redirect_stdout(io.StringIO()):
        ref_root.apply_lora_to_model(vit, ref_root.LoRAConfig(rank=4, alpha=8, dropout=0.0,
                                                              target_modules=["fc1", "fc2"]))
    names = [n for n, m in vit.named_modules() if isinstance(m, ref_root.LoRALinear)]
    with torch.no_grad():
        for n, m in vit.named_modules():
            if isinstance(m, ref_root.LoRALayer):
                m.lora_B.copy_(torch.randn(m.lora_B.shape, generator=g) * 0.05)
                out[f"lora/{n}.lora_A"] = m.lora_A.detach().numpy().copy()
                out[f"lora/{n}.lora_B"] = m.lora_B.detach().numpy().copy()
    vit.train()   # activation checkpointing path (vitdet.py:837-838); drop_path_rate is 0
    x = img.clone().requires_grad_(True)
    feat = vit(x)[0]
    (feat * gout).sum().backward()
    out["feat_lora"] = feat.detach().numpy().copy()
    out["gimg"] = x.grad.numpy().copy()
    for n, m in vit.named_modules():
        if isinstance(m, ref_root.LoRALayer):
            out[f"gA/{n}"] = m.lora_A.grad.numpy().copy()
            out[f"gB/{n}"] = m.lora_B.grad.numpy().copy()
    out["lora_module_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "vit_tiny.npz"), **out)
    print("wrote vit_tiny.npz:", len(out), "arrays;", "adapted:", names)


if __name__ == "__main__":
    main()
