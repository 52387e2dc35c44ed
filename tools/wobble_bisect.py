#!/usr/bin/env python3
"""
VERDICT r2 item 7: the reference-form adapters (plain torch autograd, oracle/lora_torch_cpu.py) run on the GPU in fp32 gave A/B
gradients that moved by 4e-3 between two evaluations of the same step ("with the allocator's state") at a bit-identical loss,
while the HIP adapters moved by 8e-7 -- which is why smoke()'s yardstick runs on the CPU.  This script bisects it:

 1. the tiny-width whole model with reference-form adapters, one training step evaluated repeatedly with the caching allocator
    perturbed in between; per adapter: x, the incoming gradient gy and the produced gA / gB are captured (hooks) -> which
    adapters move, gA or gB, and whether their INPUTS (x, gy) moved too (then the cause is upstream, in the model's backward)
    or only the outputs (then it is the adapter's own matmuls);
 2. the worst adapter's three backward products replayed stand-alone from the captured x / gy / A / B: fp32 on the GPU through
    torch.matmul under every available BLAS backend vs fp64 on the CPU, repeated under allocator perturbation.
Usage (GPU box): python tools/wobble_bisect.py > gpurun_out/wobble_bisect.txt
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

dev = "cuda:0"
cfg = dict(TINY_CONFIG, text=dict(TINY_CONFIG["text"], vocab_size=49408, context_length=32))
ds = SyntheticSegmentDataset(2, resolution=112, source=128)
batch = move_to_device(collate_fn_api([ds[0], ds[1]], dict_key="input", with_seg_masks=True)["input"], dev)
model = build_sam3_image_model(device="cpu", eval_mode=False, config=cfg, match_in_forward=False, act_checkpoint=False, seed=0)
apply_reference_form_lora(model, rank=4, alpha=8, targets=("fc1", "fc2"), only_under="vision_backbone")
ad = {n: m for n, m in model.named_modules() if isinstance(m, ReferenceFormLoRALinear)}
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for a in ad.values():
        a.lora_A.copy_(torch.randn(a.lora_A.shape, generator=g) * 0.05)
        a.lora_B.copy_(torch.randn(a.lora_B.shape, generator=g) * 0.05)
model.to(dev).train()
_, wrapper = build_criterion("local")
cap = {}
for n, m in ad.items():
    m.register_forward_hook(lambda mod, inp, out, n=n: cap.setdefault(n, {}).__setitem__("x", inp[0].detach().clone()))
    m.register_full_backward_hook(lambda mod, gin, gout, n=n: cap.setdefault(n, {}).__setitem__("gy", gout[0].detach().clone()))


def perturb(k):
    """Change what the caching allocator hands out next: free everything, then leave odd-sized holes."""
    torch.cuda.empty_cache()
    junk = [torch.empty((1 + (7 * k + 13 * i) % 29) * 257 * 1024, dtype=torch.uint8, device=dev) for i in range(k)]
    del junk[::2]
    return junk


def step():
    cap.clear()
    for a in ad.values():
        a.lora_A.grad = None
        a.lora_B.grad = None
    out = model(batch)
    targets = [model.back_convert(t) for t in batch.find_targets]
    match_all_steps(wrapper, out.output, targets)
    loss = wrapper(out, targets)["core_loss"]
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), {n: dict(cap[n], gA=a.lora_A.grad.clone(), gB=a.lora_B.grad.clone()) for n, a in ad.items()}


rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
runs = []
keep = []
for k in range(4):
    keep = perturb(3 * k)
    runs.append(step())
print("losses:", [r[0] for r in runs])
print("%-70s %10s %10s %10s %10s" % ("adapter (run k vs run 0, worst over k)", "x", "gy", "gA", "gB"))
worst = (0.0, None, None)
for n in ad:
    row = {key: max(rel(runs[k][1][n][key], runs[0][1][n][key]) for k in range(1, len(runs))) for key in ("x", "gy", "gA", "gB")}
    print("%-70s %10.2e %10.2e %10.2e %10.2e" % (n[-70:], row["x"], row["gy"], row["gA"], row["gB"]))
    for key in ("gA", "gB"):
        if row[key] > worst[0]:
            worst = (row[key], n, key)
print("worst:", worst)
n = worst[1] or next(iter(ad))
a = ad[n]
c = runs[0][1][n]
x2, gy2 = c["x"].reshape(-1, c["x"].shape[-1]), c["gy"].reshape(-1, c["gy"].shape[-1])
A, B, s = a.lora_A.detach(), a.lora_B.detach(), a.scaling
print(f"\nstand-alone replay of {n}: x {tuple(x2.shape)}, gy {tuple(gy2.shape)}, A {tuple(A.shape)}, B {tuple(B.shape)}")
xd, gd, Ad, Bd = (t.double().cpu() for t in (x2, gy2, A, B))
ref = {"gB": (xd @ Ad).t() @ (gd * s), "gt": (gd * s) @ Bd.t()}
ref["gA"] = xd.t() @ ref["gt"]
backends = ["default"]
for name in ("hipblaslt", "hipblas", "cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(name)
        backends.append(name)
    except Exception:
        pass
for be in backends:
    if be != "default":
        torch.backends.cuda.preferred_blas_library(be)
    errs = {"gB": [], "gA": [], "gt": []}
    for k in range(4):
        keep = perturb(3 * k + 1)
        gB = (x2 @ A).t() @ (gy2 * s)
        gt = (gy2 * s) @ B.t()
        gA = x2.t() @ gt
        for key, v in (("gB", gB), ("gA", gA), ("gt", gt)):
            errs[key].append(rel(v.double().cpu(), ref[key]))
    print(f"backend {be:10s}: " + "  ".join(f"{k} vs fp64 {min(v):.2e}..{max(v):.2e}" for k, v in errs.items()))
# 3. the comparison tools/rpb_diag.py made when it reported 4e-3: the decoder's position-bias KERNEL on vs the operator chain
#    (SAM3_RPB_KERNEL=1/0).  The two forms of the bias differ by ~2e-7; if A/B gradients move by 1e-3 under that, the
#    sensitivity is the model's (an ill-conditioned gradient), not a nondeterministic kernel: the table shows where it enters.
print("\nposition-bias kernel on vs off (reference-form adapters):")
res = {}
for flag in ("1", "0", "1"):
    os.environ["SAM3_RPB_KERNEL"] = flag
    res.setdefault(flag, []).append(step())
print("losses on / off / on again:", res["1"][0][0], res["0"][0][0], res["1"][1][0])
print("%-70s %10s %10s %10s %10s" % ("adapter (kernel on vs off)", "x", "gy", "gA", "gB"))
for n in ad:
    row = {key: rel(res["1"][0][1][n][key], res["0"][0][1][n][key]) for key in ("x", "gy", "gA", "gB")}
    print("%-70s %10.2e %10.2e %10.2e %10.2e" % (n[-70:], row["x"], row["gy"], row["gA"], row["gB"]))
print("on vs on again (worst gA/gB):", max(max(rel(res["1"][0][1][n][k], res["1"][1][1][n][k]) for k in ("gA", "gB")) for n in ad))
# how small are these gradients against the terms they are summed from?  |sum| / sum|.| of gA = x^T gt per adapter
print("%-70s %12s %12s" % ("adapter: cancellation of the gradient sums", "|gA|max", "sum|x||gt|max"))
for n, a in ad.items():
    c = res["1"][0][1][n]
    x2, gy2 = c["x"].reshape(-1, c["x"].shape[-1]).double(), c["gy"].reshape(-1, c["gy"].shape[-1]).double()
    gt = (gy2 * a.scaling) @ a.lora_B.detach().double().t()
    print("%-70s %12.3e %12.3e" % (n[-70:], float((x2.t() @ gt).abs().max()), float((x2.abs().t() @ gt.abs()).max())))
print("allow_tf32:", torch.backends.cuda.matmul.allow_tf32, " HIPBLASLT_ALLOW_TF32:", os.environ.get("HIPBLASLT_ALLOW_TF32"),
      " fp32_precision:", getattr(torch.backends.cuda.matmul, "fp32_precision", None))
