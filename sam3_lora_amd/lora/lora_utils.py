"""
Drop-in for ``sam3_lora.lora.lora_utils`` (reference: sam3_lora/lora/lora_utils.py,
LoRAConfig :14-56, _should_inject_lora :59-92, inject_lora_into_model :95-169,
get_lora_parameters :172-188, get_lora_state_dict :191-208, load_lora_state_dict :211-227,
merge_lora_weights :230-255, print_trainable_parameters :258-277).

Matching rule (kept bug-for-bug): a Linear is adapted when ANY target string
is a substring of its fully-qualified name -- so "proj" also hits out_proj/c_proj/hs_proj --
and the injector does NOT freeze the base model (callers do).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Set

import torch
import torch.nn as nn

from .lora_layer import LinearWithLoRA

_DEFAULT_TARGETS = ("q_proj", "k_proj", "v_proj", "out_proj", "linear1", "linear2")
_ALL_TARGETS = _DEFAULT_TARGETS + ("in_proj", "cross_attn", "self_attn")

# second matching pass of the reference: (regex on the name, pattern text the targets are tested against)
_NAME_PATTERNS = tuple((re.compile(p), p) for p in (
    r".*\.self_attn\.", r".*\.cross_attn\.", r".*\.cross_attn_image\.", r".*\.ca_text\.",
    r".*\.linear[12]$", r".*\.(q|k|v|out)_proj$"))


class LoRAConfig:
    def __init__(self, rank: int = 4, alpha: float = 1.0, dropout: float = 0.0,
                 target_modules: Optional[List[str]] = None):
        self.rank = rank
        self.alpha = alpha
        self.dropout = dropout
        targets = set(_DEFAULT_TARGETS if target_modules is None else target_modules)
        self.target_modules = set(_ALL_TARGETS) if "all" in targets else targets


def _should_inject_lora(name: str, target_modules: Set[str]) -> bool:
    if any(t in name for t in target_modules):
        return True
    # reference quirk: the fallback tests `target in <pattern text>`, not in the name
    return any(rx.match(name) and any(t in text for t in target_modules) for rx, text in _NAME_PATTERNS)


def _lora_modules(model: nn.Module):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, LinearWithLoRA)]


def inject_lora_into_model(model: nn.Module, config: LoRAConfig, verbose: bool = True) -> nn.Module:
    picked = [(n, m) for n, m in model.named_modules()
              if isinstance(m, nn.Linear) and _should_inject_lora(n, config.target_modules)]
    total_lora = 0
    for name, lin in picked:
        parent_name, _, attr = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        wrapped = LinearWithLoRA(linear=lin, rank=config.rank, alpha=config.alpha, dropout=config.dropout)
        wrapped.lora.to(device=lin.weight.device)
        setattr(parent, attr, wrapped)
        n_new = sum(p.numel() for p in wrapped.lora.parameters())
        total_lora += n_new
        if verbose:
            print(f"Injected LoRA into {name}: {lin.in_features}x{lin.out_features} -> {n_new:,} trainable params")
    if verbose:
        print(f"\nTotal LoRA injections: {len(picked)}")
        print(f"Total LoRA parameters: {total_lora:,}")
        total = sum(p.numel() for p in model.parameters())
        trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(f"Total model parameters: {total:,}")
        print(f"Trainable parameters: {trainable:,}")
        print(f"Trainable ratio: {100 * trainable / total:.2f}%")
    return model


def get_lora_parameters(model: nn.Module) -> List[nn.Parameter]:
    out: List[nn.Parameter] = []
    for _, m in _lora_modules(model):
        out.extend(m.lora.parameters())
    return out


def get_lora_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """``{"<module>.lora.lora_A": Tensor[r,in], "<module>.lora.lora_B": Tensor[out,r]}`` (plain tensors)."""
    sd = {}
    for name, m in _lora_modules(model):
        sd[f"{name}.lora.lora_A"] = m.lora.lora_A.data
        sd[f"{name}.lora.lora_B"] = m.lora.lora_B.data
    return sd


def load_lora_state_dict(model: nn.Module, state_dict: Dict[str, torch.Tensor]):
    for name, m in _lora_modules(model):
        for leaf in ("lora_A", "lora_B"):
            key = f"{name}.lora.{leaf}"
            if key in state_dict:
                getattr(m.lora, leaf).data = state_dict[key]


def merge_lora_weights(model: nn.Module) -> nn.Module:
    """Replace every LinearWithLoRA by a plain nn.Linear with the adapter folded in."""
    for name, m in _lora_modules(model):
        parent_name, _, attr = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        setattr(parent, attr, m.merge_weights())
    return model


def print_trainable_parameters(model: nn.Module):
    trainable = total = 0
    for _, p in model.named_parameters():
        total += p.numel()
        if p.requires_grad:
            trainable += p.numel()
    print(f"trainable params: {trainable:,} || all params: {total:,} || trainable%: {100 * trainable / total:.2f}")
