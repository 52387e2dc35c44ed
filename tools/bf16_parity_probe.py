#!/usr/bin/env python3
"""
Where does the bf16-layout error against the reference's fp32 run come from?  (VERDICT r2 "what's weak" 1.)

Runs the e2e fixtures (tests/golden/e2e_tiny.npz, e2e_wide.npz: the reference's whole model + loss + AdamW on CPU fp32) through
this library's model in the benchmark's layout (frozen tensors / activations bf16, A/B fp32) three times:

  hl             the product path: adapter kernels with hi + lo operands (fp32 arithmetic on the bf16 activations)
  single_round   SAM3_LORA_SINGLE_ROUND=1: A, B, t, gt rounded to bf16 once each (round 2's kernels)
  torch_fp32     the adapter branch evaluated by torch in fp32 on the same bf16 activations, autograd backward -- no adapter
                 kernel at all: what remains is the bf16 storage of activations and PyTorch-ROCm's bf16 GEMMs / attention

and prints, per variant, the element-wise error of pred_logits / pred_boxes / loss / A,B gradients / loss curve.
Usage (GPU box): python tools/bf16_parity_probe.py > gpurun_out/bf16_parity_probe.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

import numpy as np
import torch

import e2e_case_defs as D
import test_sam3_e2e as T


_ORIG = {}


def torch_fp32_adapters(model):
    """Every LoRALinear evaluates base(x) + ((x.float() @ A) @ B * s) in fp32 through torch autograd."""
    import types
    from sam3_lora_amd import functional as F_
    from sam3_lora_amd import lora_layers as L
    _ORIG.setdefault("mlp", F_.lora_mlp_gelu)
    F_.lora_mlp_gelu = lambda *a, **k: None         # module-by-module path (no fused MLP node)

    def fwd(self, x):
        base = self.original_layer(x)
        lo = self.lora
        d = (x.float() @ lo.lora_A) @ lo.lora_B * lo.scaling
        return (base.float() + d).to(base.dtype)
    for m in model.modules():
        if isinstance(m, L.LoRALinear):
            m.forward = types.MethodType(fwd, m)


ISLAND_SETS = {
    "none": (),
    "dec": ("transformer.decoder", "dot_prod_scoring"),
    "dec+enc": ("transformer.decoder", "dot_prod_scoring", "transformer.encoder"),
    "dec+enc+seg": ("transformer.decoder", "dot_prod_scoring", "transformer.encoder", "segmentation_head"),
    "all_but_trunk": ("transformer", "dot_prod_scoring", "segmentation_head", "geometry_encoder", "backbone.language_backbone",
                      "backbone.vision_backbone.convs", "backbone.vision_backbone.position_encoding"),
}


def run(which, variant, islands=None, fused=None, holes=None):
    from sam3_lora_amd import _ffi
    from sam3_lora_amd.trainer import move_to_device
    from sam3_lora_amd.vit import to_training_layout
    if "mlp" in _ORIG:
        from sam3_lora_amd import functional as F_
        F_.lora_mlp_gelu = _ORIG["mlp"]
    os.environ.pop("SAM3_LORA_SINGLE_ROUND", None)
    if variant == "single_round":
        os.environ["SAM3_LORA_SINGLE_ROUND"] = "1"
    _ffi.load().sam3_lora_debug_reload_knobs()
    gold = np.load(T.GOLD if which == "tiny" else T.GOLD_WIDE)
    dev = torch.device("cuda")
    if which == "tiny":
        model = T.build(gold, act_checkpoint=False, match_in_forward=False)
        layers, batch = T._inject(model, gold), T.make_batch()
    else:
        model = T.build_wide(gold, act_checkpoint=False, match_in_forward=False)
        layers, batch = T._inject(model, gold, D.LORA_WIDE), T.make_batch_wide()
    model.to(dev).train()
    to_training_layout(model, fp32_islands=islands, fp32_holes=holes)
    from sam3_lora_amd import functional as F2
    F2.set_fused_linear(fused)
    if variant == "torch_fp32":
        torch_fp32_adapters(model)
    m = T.run_training_steps(model, layers, gold, move_to_device(batch, dev), D.STEPS, D.CONFIGS[which][3], D.WD)
    return {"islands": list(getattr(model, "_sam3_fp32_islands", ())), "holes": len(getattr(model, "_sam3_fp32_holes", ())), "presence_logit": m["outputs"].get("presence_logit_dec"),
            "pred_masks": m["outputs"].get("pred_masks"),
            "pred_logits": max(v for k, v in m["outputs"].items() if k.endswith("pred_logits")),
            "pred_boxes": max(v for k, v in m["outputs"].items() if k.endswith("pred_boxes")),
            "queries": m["outputs"]["queries"], "encoder_hidden_states": m["outputs"]["encoder_hidden_states"],
            "core_loss": m["loss_terms"]["core_loss"], "worst_loss_term": max(m["loss_terms"].values()),
            "worst_AB_grad": max(m["grads"].values()), "loss_curve_rel": m["loss_curve_rel"], "indices_equal": m["indices_equal"]}


if __name__ == "__main__":
    res = {}
    if len(sys.argv) > 1 and sys.argv[1] == "islands":
        # which fp32 islands buy parity, with the fc1 site fused into the GEMM (default) and two-pass
        import json as _j
        yard = _j.load(open(os.path.join(ROOT, "tests", "golden", "ref_autocast_bf16.json")))
        res["reference_autocast_bf16_vs_its_fp32"] = yard
        for which in ("tiny", "wide"):
            sets = [("none", (), None), ("dec_with_bf16_holes(default)", ISLAND_SETS["dec"], None), ("dec_all_fp32", ISLAND_SETS["dec"], ()),
                    ("dec+enc+seg", ISLAND_SETS["dec+enc+seg"], None)]
            for name, isl, holes in sets:
                for rep in range(2):        # twice: the frozen GEMMs' stream-K reductions make runs differ by a few per cent
                    try:
                        res[f"{which}/{name}/run{rep}"] = run(which, "hl", islands=isl, fused=True, holes=holes)
                    except Exception as e:
                        res[f"{which}/{name}/run{rep}"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        print(json.dumps(res, indent=1))
        sys.exit(0)
    for which in ("tiny", "wide"):
        for variant in ("hl", "single_round", "torch_fp32"):
            try:
                res[f"{which}/{variant}"] = run(which, variant)
            except Exception as e:      # keep the table going
                res[f"{which}/{variant}"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    print(json.dumps(res, indent=1))
