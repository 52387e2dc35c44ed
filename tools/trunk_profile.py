#!/usr/bin/env python3
"""Where does the ViT-trunk training step spend its time?  torch.profiler kernel table (top 25)."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lora_layers as L
from sam3_lora_amd import vit as V

dev = torch.device("cuda", 0)
with torch.device(dev):
    model = V.sam3_vit()
with contextlib.redirect_stdout(io.StringIO()):
    L.apply_lora_to_model(model, L.LoRAConfig(rank=16, alpha=32, target_modules=["fc1", "fc2"], apply_to_text_encoder=False,
                                              apply_to_detr_encoder=False, apply_to_detr_decoder=False))
V.to_training_layout(model)
model.train()
img = (torch.rand(8, 3, 1008, 1008, device=dev) * 2 - 1).bfloat16()
tgt = torch.randn(8, 1024, 72, 72, device=dev).bfloat16()
def step():
    for p in L.get_lora_parameters(model):
        p.grad = None
    (model(img)[0].float() * tgt.float()).mean().backward()
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:28]
tot = sum(e.device_time_total for e in prof.key_averages())
print(f"total device time {tot/1e3:.1f} ms")
for e in rows:
    print(f"{e.device_time_total/1e3:9.2f} ms {100*e.device_time_total/tot:5.1f}% n={e.count:5d}  {e.key[:110]}")
