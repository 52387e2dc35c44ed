#!/usr/bin/env python3
"""
Which bf16 region moves the full-size outputs?  One training step of tests/golden/e2e_full.npz (depth 32, 1008^2) in the bf16
training layout with ONE more sub-module kept in fp32 at a time (vit.to_training_layout's fp32_islands), against the reference's
fp32 CPU step: logits / boxes / presence / masks / loss / worst A-B gradient / re-matched outputs and the step's peak memory.
GPU box only.   python tools/full_size_islands_probe.py > gpurun_out/full_size_islands.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

import torch

import test_sam3_e2e as T
from sam3_lora_amd.vit import DEFAULT_FP32_ISLANDS

EXTRA = {"default islands (decoder query stream + scoring head)": (),
         "+ transformer.encoder": ("transformer.encoder",),
         "+ backbone.vision_backbone.convs (neck)": ("backbone.vision_backbone.convs",),
         "+ segmentation_head": ("segmentation_head",),
         "+ backbone.language_backbone": ("backbone.language_backbone",),
         "+ backbone.vision_backbone.trunk (exact-fp32 adapters too)": ("backbone.vision_backbone.trunk",),
         "+ trunk + neck + encoder (only text tower, mask head and the decoder's memory side in bf16)":
             ("backbone.vision_backbone.trunk", "backbone.vision_backbone.convs", "transformer.encoder")}
# the HOLES inside the decoder island (vit.DEFAULT_FP32_HOLES): its memory-side work that follows bf16 by default
HOLES = {"decoder without holes (image cross-attention and position-bias MLPs in fp32 too)": (),
         "decoder holes = position-bias MLPs only (image cross-attention in fp32)": ("transformer.decoder.boxRPB_embed_x", "transformer.decoder.boxRPB_embed_y"),
         "decoder holes = image cross-attention only (position-bias MLPs in fp32)": ("transformer.decoder.layers.*.cross_attn",)}
EXTRA.update({k: None for k in HOLES})
if len(sys.argv) > 1:       # names to run (substring match)
    EXTRA = {k: v for k, v in EXTRA.items() if any(a in k for a in sys.argv[1:])}


def cast_at_every_parameterised_module(model):
    """Probe-only boundary handling for islands the library has no hooks for: every module that owns parameters casts its
    floating-point tensor inputs to the dtype of its first floating-point parameter."""
    def hook(mod, args, kwargs):
        dt = next((p.dtype for p in mod.parameters(recurse=False) if p.dtype.is_floating_point), None)
        if dt is None:
            return None
        cast = lambda t: t.to(dt) if (isinstance(t, torch.Tensor) and t.dtype.is_floating_point and t.dtype != dt) else t
        return tuple(cast(a) for a in args), {k: cast(v) for k, v in kwargs.items()}
    for m in model.modules():
        if any(True for _ in m.parameters(recurse=False)):
            m.register_forward_pre_hook(hook, with_kwargs=True)


def main():
    out = {}
    for name, extra in EXTRA.items():
        torch.cuda.reset_peak_memory_stats()
        t0 = time.time()
        try:
            if extra is None:
                rec = T._full_size_step("bf16", holes=HOLES[name])
            else:
                rec = T._full_size_step("bf16", islands=tuple(DEFAULT_FP32_ISLANDS) + tuple(extra),
                                        post_layout=cast_at_every_parameterised_module if extra else None)
            cls = lambda suffix: max(v for k, v in rec["outputs"].items() if k.endswith(suffix))
            out[name] = {"pred_logits": cls("pred_logits"), "pred_boxes": cls("pred_boxes"), "presence_logit_dec": cls("presence_logit_dec"),
                         "pred_masks": cls("pred_masks"), "core_loss": rec["loss_terms"]["core_loss"],
                         "worst_AB_grad": max(max(rec["grads_full"].values()), rec["grads_sampled_worst"]),
                         "outputs_with_different_matching": rec["outputs_with_different_matching"], "peak_mem_gb": rec["peak_mem_gb"],
                         "wall_s": round(time.time() - t0, 1)}
        except Exception as e:          # an island whose consumers have no boundary cast yet
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        print(name, out[name], file=sys.stderr, flush=True)
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
