#!/usr/bin/env python3
"""
Follow-up of tools/refform_gpu_vs_cpu.py: the gradient ARRIVING at the trunk's adapters differs between the GPU and the CPU
evaluation of the same fp32 step by 2-4e-3 of its maximum (the adapters' own backward is exact on both).  This script records,
for every leaf module of the model, the gradient w.r.t. its output and w.r.t. its input on both devices and prints the
modules whose OUTPUT gradient agrees (< 1e-5) while their INPUT gradient does not: the operator whose backward differs.
Optional: --no-miopen runs the GPU side with torch.backends.cudnn.enabled = False (ATen's own convolution kernels).
Usage (GPU box): python tools/backward_gpu_vs_cpu.py [--no-miopen] > gpurun_out/backward_gpu_vs_cpu.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from oracle.lora_torch_cpu import ReferenceFormLoRALinear, apply_reference_form_lora
from sam3_lora_amd.sam3_data import SyntheticSegmentDataset, collate_fn_api
from sam3_lora_amd.sam3_image import TINY_CONFIG, build_sam3_image_model
from sam3_lora_amd.trainer import build_criterion, match_all_steps, move_to_device

cfg = dict(TINY_CONFIG, text=dict(TINY_CONFIG["text"], vocab_size=49408, context_length=32))
ds = SyntheticSegmentDataset(2, resolution=112, source=128)


def run(where):
    batch = move_to_device(collate_fn_api([ds[0], ds[1]], dict_key="input", with_seg_masks=True)["input"], where)
    model = build_sam3_image_model(device="cpu", eval_mode=False, config=cfg, match_in_forward=False, act_checkpoint=False, seed=0)
    apply_reference_form_lora(model, rank=4, alpha=8, targets=("fc1", "fc2"), only_under="vision_backbone")
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for a in (m for m in model.modules() if isinstance(m, ReferenceFormLoRALinear)):
            a.lora_A.copy_(torch.randn(a.lora_A.shape, generator=g) * 0.05)
            a.lora_B.copy_(torch.randn(a.lora_B.shape, generator=g) * 0.05)
    model.to(where).train()
    _, wrapper = build_criterion("local")
    model.set_prefetch_matcher(wrapper)
    cap, order = {}, []

    def hook(n):
        def f(mod, gin, gout):
            if n in cap:
                return
            order.append(n)
            first = lambda ts: next((t.detach().double().cpu() for t in ts if isinstance(t, torch.Tensor)), None)
            cap[n] = (type(mod).__name__, first(gout), first(gin))
        return f
    fwd = {}
    for n, m in model.named_modules():
        if not list(m.children()):
            m.register_full_backward_hook(hook(n))
            if n.endswith("linear1"):
                def keep(mod, inp, out, n=n):
                    fwd.setdefault(n, out.detach().double().cpu())        # (returns None: the output stays as it is)
                m.register_forward_hook(keep)
    out = model(batch)
    targets = [model.back_convert(t) for t in batch.find_targets]
    match_all_steps(wrapper, out.output, targets)
    loss = wrapper(out, targets)["core_loss"]
    loss.backward()
    return loss.item(), cap, order, fwd


if "--no-miopen" in sys.argv:
    torch.backends.cudnn.enabled = False
rel = lambda a, b: (float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
                    if a is not None and b is not None and a.shape == b.shape and a.numel() else float("nan"))
lc, c, _, fc = run("cpu")
lg, g, order, fg = run("cuda:0")
print(f"loss cpu {lc:.9f} gpu {lg:.9f}   cudnn(MIOpen).enabled={torch.backends.cudnn.enabled}")
print("modules in the order their backward ran on the GPU; shown: output gradient agrees (< 1e-5) but input gradient does not (> 1e-4)")
bad = 0
for n in order:
    if n not in c:
        continue
    kind, go, gi = g[n]
    eo, ei = rel(go, c[n][1]), rel(gi, c[n][2])
    if eo == eo and ei == ei and eo < 1e-5 and ei > 1e-4:
        bad += 1
        print("%-34s %-80s gout %.1e  gin %.1e  %s" % (kind, n[-80:], eo, ei, tuple(go.shape)))
print(f"{bad} such modules")
print("\nfirst 25 modules (backward order) with their output / input gradient differences:")
for n in order[:25]:
    if n in c:
        print("%-30s %-70s gout %.1e gin %.1e" % (g[n][0], n[-70:], rel(g[n][1], c[n][1]), rel(g[n][2], c[n][2])))

print("\nmodules (backward order) around the first output gradient that differs by > 1e-4:")
errs = [(n, rel(g[n][1], c[n][1]), rel(g[n][2], c[n][2])) for n in order if n in c]
first = next((i for i, e in enumerate(errs) if e[1] == e[1] and e[1] > 1e-4), None)
if first is not None:
    for n, eo, ei in errs[max(0, first - 12):first + 12]:
        print("%-30s %-80s gout %.1e gin %.1e" % (g[n][0], n[-80:], eo, ei))

print("\nReLU gates of the FFNs (pre-activation = output of linear1): elements whose SIGN differs between the devices")
for n in fc:
    a, b = fg[n], fc[n]
    flip = (a > 0) != (b > 0)
    k = int(flip.sum())
    mag = float(torch.maximum(a.abs(), b.abs())[flip].max()) if k else 0.0
    print("%-60s %8d elements  %3d gate flips  largest |pre-activation| among them %.2e   max |pre-activation| %.2e   forward diff %.1e"
          % (n[-60:], a.numel(), k, mag, float(b.abs().max()), rel(a, b)))
