"""
Harness (build container only) that instantiates the REFERENCE SAM3 image model on the meta
device, with stand-ins for third-party packages the image lacks, and records

  * every nn.Linear of the model: (qualified name, in_features, out_features, has_bias)
  * which of them each reference injector adapts, for every shipped YAML config and for
    a few package-injector target sets

into ``sam3_linears.json``.  Nothing of the reference's source is stored -- only names and
integers.  Invoked via ``make_golden.py --sam3``.
"""
import contextlib
import glob
import importlib.abc
import importlib.machinery
import io
import json
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


# ---------------------------------------------------------------- permissive stand-ins --
class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any()

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {"__init__": lambda self, *a, **kw: None,
                            "__call__": lambda self, *a, **kw: None})


STUB_ROOTS = ("timm", "torchvision", "iopath", "pycocotools", "ftfy", "torchmetrics", "cv2",
              "decord", "hydra", "omegaconf", "submitit", "fvcore", "matplotlib", "skimage",
              "pandas_stub_never", "scipy_stub_never", "triton_stub_never", "open_clip", "numba")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def _install_stubs():
    sys.meta_path.insert(0, _Finder())
    # timm.layers needs real structure: the ViT MLP's fc1/fc2 names come from timm's Mlp
    tl = _StubModule("timm.layers")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return x

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                     norm_layer=None, bias=True, drop=0.0, use_conv=False):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            drop = drop if isinstance(drop, (tuple, list)) else (drop, drop)
            self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop[0])
            self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
            self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
            self.drop2 = nn.Dropout(drop[1])

        def forward(self, x):
            return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))

    tl.DropPath = DropPath
    tl.Mlp = Mlp
    tl.trunc_normal_ = lambda t, *a, **k: t
    sys.modules["timm.layers"] = tl
    tv_ops = _StubModule("torchvision.ops")

    class RoIAlign(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tv_ops.RoIAlign = RoIAlign
    tv_ops.roi_align = lambda *a, **k: None
    sys.modules["torchvision.ops"] = tv_ops
    sys.modules["torchvision.ops.roi_align"] = tv_ops
    iop = _StubModule("iopath.common.file_io")

    class _PM:
        def open(self, p, mode="r", **k):
            return open(p, mode)

        def exists(self, p):
            return os.path.exists(p)

    iop.g_pathmgr = _PM()
    sys.modules["iopath.common.file_io"] = iop


def _patch_cuda_literals():
    """position_encoding.py:47 and decoder.py:281 pass device="cuda" literally."""
    def wrap(fn):
        def inner(*a, **k):
            if k.get("device") == "cuda":
                k.pop("device")
            return fn(*a, **k)
        return inner
    for name in ("zeros", "arange", "ones", "empty", "tensor", "linspace"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.pin_memory = lambda self, *a, **k: self


def build_reference_model_meta():
    sys.path.insert(0, REF)
    _install_stubs()
    _patch_cuda_literals()
    from sam3.model_builder import build_sam3_image_model
    with torch.device("meta"):
        model = build_sam3_image_model(
            device="meta", eval_mode=False, checkpoint_path=None, load_from_HF=False,
            bpe_path=os.path.join(REF, "sam3", "assets", "bpe_simple_vocab_16e6.txt.gz"))
    return model


def main(out_dir, ref_root, ref_pkg_utils, ref_pkg_layer):
    import copy
    import yaml
    model = build_reference_model_meta()
    linears = [(n, m.in_features, m.out_features, m.bias is not None)
               for n, m in model.named_modules() if isinstance(m, nn.Linear)]
    total = sum(p.numel() for p in model.parameters())
    out = dict(total_parameters=int(total), linears=linears, root={}, package={})
    for path in sorted(glob.glob(os.path.join(REF, "configs", "*.yaml"))):
        cfg = yaml.safe_load(open(path))
        lc = cfg.get("lora")
        if not lc or "apply_to_vision_encoder" not in lc:
            continue
        m = copy.deepcopy(model)
        keys = ("rank", "alpha", "dropout", "target_modules", "apply_to_vision_encoder",
                "apply_to_text_encoder", "apply_to_geometry_encoder", "apply_to_detr_encoder",
                "apply_to_detr_decoder", "apply_to_mask_decoder")
        conf = ref_root.LoRAConfig(**{k: lc[k] for k in keys})
        with contextlib.redirect_stdout(io.StringIO()), torch.device("meta"):
            ref_root.apply_lora_to_model(m, conf)
        names = [n for n, mm in m.named_modules() if isinstance(mm, ref_root.LoRALinear)]
        cnt = ref_root.count_parameters(m)
        out["root"][os.path.basename(path)] = dict(
            lora={k: lc[k] for k in keys}, names=names,
            trainable_parameters=int(cnt["trainable_parameters"]),
            total_parameters=int(cnt["total_parameters"]))
        print("root", os.path.basename(path), len(names), cnt["trainable_parameters"])
    for key, (tm, r) in {"default_r16": (None, 16), "fc_r16": (["fc1", "fc2"], 16),
                         "qkv_proj_r16": (["qkv", "proj"], 16), "all_r8": (["all"], 8)}.items():
        m = copy.deepcopy(model)
        with contextlib.redirect_stdout(io.StringIO()), torch.device("meta"):
            ref_pkg_utils.inject_lora_into_model(m, ref_pkg_utils.LoRAConfig(rank=r, alpha=2.0 * r, target_modules=tm), False)
        names = [n for n, mm in m.named_modules() if isinstance(mm, ref_pkg_layer.LinearWithLoRA)]
        n_el = int(sum(p.numel() for p in ref_pkg_utils.get_lora_parameters(m)))
        out["package"][key] = dict(target_modules=tm, rank=r, names=names, n_lora_elems=n_el)
        print("package", key, len(names), n_el)
    json.dump(out, open(os.path.join(out_dir, "sam3_linears.json"), "w"))
    print("linears:", len(linears), "total params:", total)
