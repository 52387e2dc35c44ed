#!/usr/bin/env python3
"""
smoke()'s third leg: the reference-form adapters (plain torch autograd, fp32, oracle/lora_torch_cpu.py) evaluated on the GPU
differ from the same on the CPU by 4e-3 in an A/B gradient at an identical loss, while the HIP adapters on the GPU agree
with the CPU to 2e-6.  tools/wobble_bisect.py showed the GPU side is bit-reproducible run to run; this script compares the
two devices adapter by adapter -- x, incoming gradient gy, gA, gB -- and checks each device's gA / gB against an fp64
evaluation of the adapter's own backward on the x / gy THAT device captured (is the adapter's backward wrong, or its inputs?).
Usage (GPU box): python tools/refform_gpu_vs_cpu.py > gpurun_out/refform_gpu_vs_cpu.txt
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
    ad = {n: m for n, m in model.named_modules() if isinstance(m, ReferenceFormLoRALinear)}
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for a in ad.values():
            a.lora_A.copy_(torch.randn(a.lora_A.shape, generator=g) * 0.05)
            a.lora_B.copy_(torch.randn(a.lora_B.shape, generator=g) * 0.05)
    model.to(where).train()
    _, wrapper = build_criterion("local")
    model.set_prefetch_matcher(wrapper)
    cap = {}
    for n, m in ad.items():
        m.register_forward_hook(lambda mod, inp, out, n=n: cap.setdefault(n, {}).__setitem__("x", inp[0].detach().clone()))
        m.register_full_backward_hook(lambda mod, gin, gout, n=n: cap.setdefault(n, {}).__setitem__("gy", gout[0].detach().clone()))
    out = model(batch)
    targets = [model.back_convert(t) for t in batch.find_targets]
    match_all_steps(wrapper, out.output, targets)
    loss = wrapper(out, targets)["core_loss"]
    loss.backward()
    if where != "cpu":
        torch.cuda.synchronize()
    res = {}
    for n, a in ad.items():
        x2 = cap[n]["x"].reshape(-1, cap[n]["x"].shape[-1]).double().cpu()
        gy2 = cap[n]["gy"].reshape(-1, cap[n]["gy"].shape[-1]).double().cpu()
        A, B, s = a.lora_A.detach().double().cpu(), a.lora_B.detach().double().cpu(), a.scaling
        gt = (gy2 * s) @ B.t()
        res[n] = dict(x=x2, gy=gy2, gA=a.lora_A.grad.double().cpu(), gB=a.lora_B.grad.double().cpu(),
                      gA64=x2.t() @ gt, gB64=(x2 @ A).t() @ (gy2 * s), absA=(x2.abs().t() @ gt.abs()))
    return float(loss), res


rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
lc, c = run("cpu")
lg, g = run("cuda:0")
print(f"loss cpu {lc:.9f} gpu {lg:.9f}")
print("%-58s %9s %9s %9s %9s | %9s %9s | %9s %9s | %9s" % ("adapter: GPU vs CPU", "x", "gy", "gA", "gB", "gpu gA/64", "gpu gB/64",
                                                         "cpu gA/64", "cpu gB/64", "cancel gA"))
for n in c:
    print("%-58s %9.1e %9.1e %9.1e %9.1e | %9.1e %9.1e | %9.1e %9.1e | %9.1e" % (
        n[-58:], rel(g[n]["x"], c[n]["x"]), rel(g[n]["gy"], c[n]["gy"]), rel(g[n]["gA"], c[n]["gA"]), rel(g[n]["gB"], c[n]["gB"]),
        rel(g[n]["gA"], g[n]["gA64"]), rel(g[n]["gB"], g[n]["gB64"]), rel(c[n]["gA"], c[n]["gA64"]), rel(c[n]["gB"], c[n]["gB64"]),
        float(c[n]["gA64"].abs().max() / c[n]["absA"].max())))
# where along the rows does gy differ?  (padding tokens / windows)
worst = max(c, key=lambda n: rel(g[n]["gy"], c[n]["gy"]))
d = (g[worst]["gy"] - c[worst]["gy"]).abs().max(dim=1).values
print(f"\nworst gy: {worst}: rows with |diff| > 1e-3 of max: {(d > 1e-3 * c[worst]['gy'].abs().max()).nonzero().flatten().tolist()[:40]} of {d.numel()}")
print("gy row norms (cpu) at those rows vs median:", c[worst]["gy"].norm(dim=1)[d.argmax()].item(), c[worst]["gy"].norm(dim=1).median().item())
