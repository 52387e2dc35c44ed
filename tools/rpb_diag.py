"""Why smoke()'s oracle side runs on the CPU: the whole tiny-model step with the adapters in HIP form and in reference form
(plain torch autograd), both on the GPU, with the decoder's position-bias kernel on and off.  Finding (MI355X, ROCm 7.2,
torch 2.10): the bias agrees to 2.4e-7 and every forward output to < 1e-6 either way; the HIP form's A/B gradients move by
7.9e-7 -- the GPU reference form's by 4.2e-3 with a bit-identical loss: some fp32 library kernel in ITS backward changes
precision with the allocator's state.  Against the CPU fp32 reference form both settings agree to 2e-6."""
import os, sys, io, contextlib
sys.path.insert(0, os.getcwd())
import torch
import lora_layers as L
from oracle.lora_torch_cpu import ReferenceFormLoRALinear, apply_reference_form_lora
from sam3_lora_amd.sam3_data import SyntheticSegmentDataset, collate_fn_api
from sam3_lora_amd.sam3_image import TINY_CONFIG, build_sam3_image_model
from sam3_lora_amd.trainer import build_criterion, match_all_steps, move_to_device
dev = "cuda:0"
cfg = dict(TINY_CONFIG, text=dict(TINY_CONFIG["text"], vocab_size=49408, context_length=32))
ds = SyntheticSegmentDataset(2, resolution=112, source=128)
batch = move_to_device(collate_fn_api([ds[0], ds[1]], dict_key="input", with_seg_masks=True)["input"], dev)
def build(form):
    model = build_sam3_image_model(device="cpu", eval_mode=False, config=cfg, match_in_forward=False, act_checkpoint=False, seed=0)
    if form == "hip":
        with contextlib.redirect_stdout(io.StringIO()):
            L.apply_lora_to_model(model, L.LoRAConfig(rank=4, alpha=8, dropout=0.0, target_modules=["fc1", "fc2"], apply_to_vision_encoder=True, apply_to_text_encoder=False, apply_to_geometry_encoder=False, apply_to_detr_encoder=False, apply_to_detr_decoder=False, apply_to_mask_decoder=False))
        ad = [m.lora for m in model.modules() if isinstance(m, L.LoRALinear)]
    else:
        apply_reference_form_lora(model, rank=4, alpha=8, targets=("fc1", "fc2"), only_under="vision_backbone")
        ad = [m for m in model.modules() if isinstance(m, ReferenceFormLoRALinear)]
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for a in ad:
            a.lora_A.copy_(torch.randn(a.lora_A.shape, generator=g) * 0.05)
            a.lora_B.copy_(torch.randn(a.lora_B.shape, generator=g) * 0.05)
    return model.to(dev).train(), ad
def step(model, ad, kernel):
    os.environ["SAM3_RPB_KERNEL"] = "1" if kernel else "0"
    for a in ad:
        a.lora_A.grad = None; a.lora_B.grad = None
    _, wrapper = build_criterion("local")
    out = model(batch)
    targets = [model.back_convert(t) for t in batch.find_targets]
    match_all_steps(wrapper, out.output, targets)
    loss = wrapper(out, targets)["core_loss"]
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), [a.lora_A.grad.clone() for a in ad] + [a.lora_B.grad.clone() for a in ad], out.output[0][0]
rel = lambda a, b: max(float((x - y).abs().max() / y.abs().max().clamp_min(1e-12)) for x, y in zip(a, b))
res = {}
for form in ("hip", "reference"):
    m, ad = build(form)
    for k in (False, True):
        res[(form, k)] = step(m, ad, k)
    res[(form, "again")] = step(m, ad, True)
for form in ("hip", "reference"):
    print(form, "loss kernel off / on / on again:", res[(form, False)][0], res[(form, True)][0], res[(form, "again")][0],
          " grads on vs off: %.2e  on vs on-again: %.2e" % (rel(res[(form, True)][1], res[(form, False)][1]), rel(res[(form, True)][1], res[(form, "again")][1])))
for k in (False, True):
    print("kernel", k, ": hip vs reference grads %.2e" % rel(res[("hip", k)][1], res[("reference", k)][1]))
o_on, o_off = res[("hip", True)][2], res[("hip", False)][2]
for key in ("pred_logits", "pred_boxes", "pred_masks"):
    print(key, "hip on vs off: %.2e" % float((o_on[key].float() - o_off[key].float()).abs().max() / o_off[key].float().abs().max()))
# the bias itself on the decoder's actual first-layer boxes
dec = m.transformer.decoder
rb = dec.reference_points.weight.float().unsqueeze(1).repeat(2, 2, 1).sigmoid()
os.environ["SAM3_RPB_KERNEL"] = "1"; a = dec._get_rpb_matrix(rb, (8, 8), presence_row=True)
os.environ["SAM3_RPB_KERNEL"] = "0"; b = dec._get_rpb_matrix(rb, (8, 8), presence_row=True)
print("bias kernel vs operators: shape", tuple(a.shape), tuple(b.shape), "max abs diff %.3e of max %.3e" % (float((a - b).abs().max()), float(b.abs().max())))
