#!/usr/bin/env python3
"""Where does the HOST spend its time in a batch-1 trunk step (launch-bound regime)?  cProfile, top by own time."""
import contextlib, cProfile, io, os, pstats, sys
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
img = (torch.rand(B, 3, 1008, 1008, device=dev) * 2 - 1).bfloat16()
tgt = torch.randn(B, 1024, 72, 72, device=dev).bfloat16()
def step():
    for p in L.get_lora_parameters(model):
        p.grad = None
    (model(img)[0].float() * tgt.float()).mean().backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(28)
