"""
Drop-in for the reference's top-level ``lora_layers`` module (the API the north-star CLI
``train_sam3_lora_native.py`` imports), with the adapter arithmetic on the MI355X HIP path.

Same public names, argument names, defaults, parameter shapes/init, state-dict keys and
freezing behaviour as the reference (``lora_layers.py``: LoRALayer :13-55, LoRALinear :58-91,
LoRAConfig :94-155, apply_lora_to_model :158-228, get_lora_parameters :231-245,
count_parameters :248-262, save_lora_weights :265-280, load_lora_weights :283-293).

Parameter layout ("root"):  lora_A [in_features, rank], lora_B [rank, out_features].
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ._ffi import MAX_RANK
from .functional import LAYOUT_ROOT, PackedOperands, TransposedCopy, lora_linear

__all__ = [
    "LoRALayer", "LoRALinear", "LoRAConfig", "apply_lora_to_model", "get_lora_parameters",
    "count_parameters", "save_lora_weights", "load_lora_weights",
]


class LoRALayer(nn.Module):
    """Low-rank branch ``(dropout(x) @ lora_A @ lora_B) * (alpha / rank)``.

    Inside a ``LoRALinear`` the branch is never evaluated on its own: the wrapper fuses it
    with the frozen layer's output.  Calling the layer directly still works (it runs the same
    HIP kernels against a zero base) so code that composes ``base(x) + lora(x)`` by hand keeps
    its meaning.
    """

    def __init__(self, in_features: int, out_features: int, rank: int = 8, alpha: int = 16,
                 dropout: float = 0.0):
        super().__init__()
        if not 1 <= int(rank) <= MAX_RANK:
            raise ValueError(f"LoRA rank must be in [1, {MAX_RANK}] (got {rank})")
        self.rank = rank
        self.alpha = alpha
        self.scaling = alpha / rank
        a = torch.empty(in_features, rank)
        # kaiming_uniform_(a=sqrt(5)): bound = 1/sqrt(fan_in) with fan_in = size(1) = rank
        nn.init.kaiming_uniform_(a, a=math.sqrt(5))
        self.lora_A = nn.Parameter(a)
        self.lora_B = nn.Parameter(torch.zeros(rank, out_features))
        self.dropout = nn.Dropout(p=dropout) if dropout > 0 else nn.Identity()
        self._packed = PackedOperands()   # bf16 operand images of A/B, re-packed when they change (not state)

    @property
    def dropout_p(self) -> float:
        return float(self.dropout.p) if isinstance(self.dropout, nn.Dropout) else 0.0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return lora_linear(x, None, None, self.lora_A, self.lora_B, self.scaling, LAYOUT_ROOT,
                           self.dropout_p, self.training, cache=self._packed)


class LoRALinear(nn.Module):
    """Frozen ``nn.Linear`` + LoRA branch: ``original_layer(x) + lora(x)`` as ONE fused op."""

    def __init__(self, original_layer: nn.Linear, rank: int = 8, alpha: int = 16, dropout: float = 0.0):
        super().__init__()
        self.original_layer = original_layer
        for p in self.original_layer.parameters():
            p.requires_grad = False
        self.lora = LoRALayer(original_layer.in_features, original_layer.out_features,
                              rank=rank, alpha=alpha, dropout=dropout)
        self._wt = TransposedCopy()       # W^T of the frozen weight for the TN-form input-gradient GEMM (not state)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lo = self.lora
        return lora_linear(x, self.original_layer.weight, self.original_layer.bias, lo.lora_A, lo.lora_B,
                           lo.scaling, LAYOUT_ROOT, lo.dropout_p, self.training, cache=lo._packed, wt_cache=self._wt)


class LoRAConfig:
    """Which modules get an adapter, and with what rank/alpha/dropout."""

    _FLAGS = ("apply_to_vision_encoder", "apply_to_text_encoder", "apply_to_geometry_encoder",
              "apply_to_detr_encoder", "apply_to_detr_decoder", "apply_to_mask_decoder")

    def __init__(self, rank: int = 8, alpha: int = 16, dropout: float = 0.0,
                 target_modules: Optional[List[str]] = None,
                 apply_to_vision_encoder: bool = True, apply_to_text_encoder: bool = True,
                 apply_to_geometry_encoder: bool = False, apply_to_detr_encoder: bool = True,
                 apply_to_detr_decoder: bool = True, apply_to_mask_decoder: bool = False):
        self.rank = rank
        self.alpha = alpha
        self.dropout = dropout
        self.target_modules = set(["q_proj", "k_proj", "v_proj", "out_proj"]
                                  if target_modules is None else target_modules)
        self.apply_to_vision_encoder = apply_to_vision_encoder
        self.apply_to_text_encoder = apply_to_text_encoder
        self.apply_to_geometry_encoder = apply_to_geometry_encoder
        self.apply_to_detr_encoder = apply_to_detr_encoder
        self.apply_to_detr_decoder = apply_to_detr_decoder
        self.apply_to_mask_decoder = apply_to_mask_decoder

    def to_dict(self) -> Dict:
        d = {"rank": self.rank, "alpha": self.alpha, "dropout": self.dropout,
             "target_modules": list(self.target_modules)}
        d.update({f: getattr(self, f) for f in self._FLAGS})
        return d


# component gate: (substrings that identify the component, config flag)
_COMPONENTS = (
    (("vision_encoder", "vision_backbone"), "apply_to_vision_encoder"),
    (("text_encoder", "language_backbone"), "apply_to_text_encoder"),
    (("geometry_encoder",), "apply_to_geometry_encoder"),
    (("detr_encoder", "transformer.encoder"), "apply_to_detr_encoder"),
    (("detr_decoder", "transformer.decoder"), "apply_to_detr_decoder"),
    (("mask_decoder",), "apply_to_mask_decoder"),
)


def _wants_adapter(qualified_name: str, config: LoRAConfig) -> bool:
    for needles, flag in _COMPONENTS:
        if not getattr(config, flag) and any(s in qualified_name for s in needles):
            return False
    leaf = qualified_name.rsplit(".", 1)[-1]
    # nn.MultiheadAttention reads out_proj.weight directly, so out_proj is never wrapped
    return leaf != "out_proj" and leaf in config.target_modules


def apply_lora_to_model(model: nn.Module, config: LoRAConfig) -> nn.Module:
    """Freeze every parameter, then wrap each matching ``nn.Linear`` in a ``LoRALinear`` (in place)."""
    for p in model.parameters():
        p.requires_grad = False
    chosen = [(name, mod) for name, mod in model.named_modules()
              if isinstance(mod, nn.Linear) and _wants_adapter(name, config)]
    for name, mod in chosen:
        parent_name, _, attr = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        wrapped = LoRALinear(mod, rank=config.rank, alpha=config.alpha, dropout=config.dropout)
        wrapped.lora.to(device=mod.weight.device)
        setattr(parent, attr, wrapped)
    print(f"Applied LoRA to {len(chosen)} modules:")
    for name, _ in chosen[:10]:
        print(f"  - {name}")
    if len(chosen) > 10:
        print(f"  ... and {len(chosen) - 10} more")
    return model


def get_lora_parameters(model: nn.Module) -> List[nn.Parameter]:
    out: List[nn.Parameter] = []
    for m in model.modules():
        if isinstance(m, LoRALayer):
            out += [m.lora_A, m.lora_B]
    return out


def count_parameters(model: nn.Module) -> Dict[str, int]:
    total = trainable = 0
    for p in model.parameters():
        total += p.numel()
        if p.requires_grad:
            trainable += p.numel()
    return {"total_parameters": total, "trainable_parameters": trainable,
            "trainable_percentage": 100 * trainable / total if total > 0 else 0}


def save_lora_weights(model: nn.Module, save_path: str):
    """``{"<LoRALayer path>.lora_A": Parameter[in,r], "<...>.lora_B": Parameter[r,out]}`` via torch.save."""
    blob = {}
    for name, m in model.named_modules():
        if isinstance(m, LoRALayer):
            blob[f"{name}.lora_A"] = m.lora_A
            blob[f"{name}.lora_B"] = m.lora_B
    torch.save(blob, save_path)
    print(f"Saved LoRA weights to {save_path}")


def load_lora_weights(model: nn.Module, load_path: str):
    blob = torch.load(load_path, map_location="cpu")    # default (safe) unpickler: the file holds tensors only
    model.load_state_dict(blob, strict=False)
    print(f"Loaded LoRA weights from {load_path}")
