#!/usr/bin/env python3
"""
Generate the golden fixtures in this directory by RUNNING THE REFERENCE (imported from
/root/reference, CPU, fp32) on the seeded inputs of ``cases.py``.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # adapter vectors + toy-model manifests
    python tests/golden/make_golden.py --sam3     # additionally: SAM3 Linear-name manifest
                                                  # (builds the 840M-param reference model
                                                  # with harness-side stubs; ~2-5 min)

Outputs (data only -- no reference source is stored):
    adapter_<case>.npz     y, gx, gA, gB (fp32) of the reference's LoRALinear / LinearWithLoRA
                           forward + autograd backward
    init_stats.json        empirical init bounds of lora_A / zeros of lora_B
    toy_manifests.json     module names the reference injectors pick on small models
    ckpt_keys.json         state-dict key lists of both checkpoint formats
    sam3_linears.json      (--sam3) every nn.Linear of the reference SAM3 image model
                           (name, in, out) + the names each reference injector/config adapts
"""
import argparse
import io
import contextlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
# the reference must win over this repo's same-named drop-in shims
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

import cases  # noqa: E402

import lora_layers as ref_root  # noqa: E402  (reference)
from sam3_lora.lora import lora_layer as ref_pkg_layer  # noqa: E402
from sam3_lora.lora import lora_utils as ref_pkg_utils  # noqa: E402

assert ref_root.__file__.startswith(REF), ref_root.__file__
assert ref_pkg_layer.__file__.startswith(REF), ref_pkg_layer.__file__


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def run_adapter_case(name):
    c = cases.make_case(name)
    fin, fout = c["W"].shape[1], c["W"].shape[0]
    lin = nn.Linear(fin, fout, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(c["W"]))
        lin.bias.copy_(torch.from_numpy(c["b"]))
    if c["layout"] == cases.LAYOUT_ROOT:
        mod = ref_root.LoRALinear(lin, rank=c["rank"], alpha=c["alpha"], dropout=0.0)
    else:
        mod = ref_pkg_layer.LinearWithLoRA(lin, rank=c["rank"], alpha=c["alpha"], dropout=0.0)
    with torch.no_grad():
        mod.lora.lora_A.copy_(torch.from_numpy(c["A"]))
        mod.lora.lora_B.copy_(torch.from_numpy(c["B"]))
    x = torch.from_numpy(c["x"]).clone().requires_grad_(True)
    y = mod(x)
    y.backward(torch.from_numpy(c["gy"]))
    out = dict(
        y=y.detach().numpy(), gx=x.grad.numpy(),
        gA=mod.lora.lora_A.grad.numpy(), gB=mod.lora.lora_B.grad.numpy(),
    )
    if c["layout"] == cases.LAYOUT_PACKAGE:
        out["merged_weight"] = mod.merge_weights().weight.detach().numpy() if fin * fout <= 140_000 else np.zeros(0, np.float32)
    np.savez_compressed(os.path.join(HERE, f"adapter_{name}.npz"), **out)
    return {k: list(v.shape) for k, v in out.items()}


def init_stats():
    torch.manual_seed(0)
    res = {}
    for (fin, fout, r) in [(1024, 4736, 16), (4736, 1024, 16), (256, 2048, 4), (1024, 4736, 32)]:
        a = ref_root.LoRALayer(fin, fout, rank=r, alpha=2 * r)
        p = ref_pkg_layer.LoRALayer(fin, fout, rank=r, alpha=2.0 * r)
        res[f"{fin}x{fout}_r{r}"] = dict(
            root_A_shape=list(a.lora_A.shape), root_B_shape=list(a.lora_B.shape),
            root_A_absmax=float(a.lora_A.detach().abs().max()), root_B_absmax=float(a.lora_B.detach().abs().max()),
            root_scaling=float(a.scaling),
            pkg_A_shape=list(p.lora_A.shape), pkg_B_shape=list(p.lora_B.shape),
            pkg_A_absmax=float(p.lora_A.detach().abs().max()), pkg_B_absmax=float(p.lora_B.detach().abs().max()),
            pkg_scaling=float(p.scaling),
        )
    return res


from make_golden_models import ToySam, ROOT_CONFIGS, PKG_CONFIGS  # noqa: E402


def lora_names_root(model):
    return [n for n, m in model.named_modules() if isinstance(m, ref_root.LoRALinear)]


def lora_names_pkg(model):
    return [n for n, m in model.named_modules() if isinstance(m, ref_pkg_layer.LinearWithLoRA)]


def toy_manifests():
    out = {"root": {}, "package": {}}
    for k, kw in ROOT_CONFIGS.items():
        torch.manual_seed(0)
        m = ToySam()
        quiet(ref_root.apply_lora_to_model, m, ref_root.LoRAConfig(rank=4, alpha=8, **kw))
        cnt = ref_root.count_parameters(m)
        out["root"][k] = dict(names=lora_names_root(m), counts=cnt,
                              n_lora_params=len(ref_root.get_lora_parameters(m)),
                              any_base_trainable=any(p.requires_grad for n, p in m.named_parameters() if "lora_" not in n))
    for k, tm in PKG_CONFIGS.items():
        torch.manual_seed(0)
        m = ToySam()
        quiet(ref_pkg_utils.inject_lora_into_model, m, ref_pkg_utils.LoRAConfig(rank=4, alpha=8.0, target_modules=tm), False)
        out["package"][k] = dict(names=lora_names_pkg(m),
                                 n_lora_params=len(ref_pkg_utils.get_lora_parameters(m)),
                                 n_lora_elems=int(sum(p.numel() for p in ref_pkg_utils.get_lora_parameters(m))),
                                 any_base_trainable=any(p.requires_grad for n, p in m.named_parameters() if "lora_" not in n))
    return out


def ckpt_keys():
    torch.manual_seed(0)
    m = ToySam()
    quiet(ref_root.apply_lora_to_model, m, ref_root.LoRAConfig(rank=4, alpha=8, **ROOT_CONFIGS["fc_only_vision"]))
    buf = io.BytesIO()
    with contextlib.redirect_stdout(io.StringIO()):
        # save_lora_weights writes {"<LoRALayer name>.lora_A": Parameter, ...} (lora_layers.py:265-280)
        path = os.path.join(HERE, "_tmp_root.pt")
        ref_root.save_lora_weights(m, path)
    sd = torch.load(path, weights_only=False)
    os.remove(path)
    root = {k: dict(shape=list(v.shape), is_parameter=isinstance(v, nn.Parameter)) for k, v in sd.items()}
    m2 = ToySam()
    quiet(ref_pkg_utils.inject_lora_into_model, m2, ref_pkg_utils.LoRAConfig(rank=4, alpha=8.0, target_modules=["fc1", "fc2"]), False)
    psd = ref_pkg_utils.get_lora_state_dict(m2)
    pkg = {k: dict(shape=list(v.shape), is_parameter=isinstance(v, nn.Parameter)) for k, v in psd.items()}
    return dict(root=root, package=pkg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sam3", action="store_true")
    args = ap.parse_args()
    shapes = {}
    for name in cases.CASES:
        shapes[name] = run_adapter_case(name)
        print("adapter", name, shapes[name])
    json.dump(init_stats(), open(os.path.join(HERE, "init_stats.json"), "w"), indent=1)
    json.dump(toy_manifests(), open(os.path.join(HERE, "toy_manifests.json"), "w"), indent=1)
    json.dump(ckpt_keys(), open(os.path.join(HERE, "ckpt_keys.json"), "w"), indent=1)
    if args.sam3:
        import sam3_manifest
        sam3_manifest.main(HERE, ref_root, ref_pkg_utils, ref_pkg_layer)


if __name__ == "__main__":
    main()
