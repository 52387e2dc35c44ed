"""
Drop-in for ``sam3_lora.lora.lora_layer`` (reference: sam3_lora/lora/lora_layer.py,
LoRALayer :16-88, LinearWithLoRA :91-178) on the MI355X HIP path.

Parameter layout ("package"): lora_A [rank, in_features], lora_B [out_features, rank].
The reference materialises ``lora_B @ lora_A`` ([out, in]) on every call and runs a second
dense GEMM; here the branch is two rank-r contractions fused into the frozen layer's output.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .._ffi import MAX_RANK
from ..functional import LAYOUT_PACKAGE, PackedOperands, TransposedCopy, lora_linear, merge_weight


class LoRALayer(nn.Module):
    def __init__(self, in_features: int, out_features: int, rank: int = 4, alpha: float = 1.0,
                 dropout: float = 0.0):
        super().__init__()
        if not 1 <= int(rank) <= MAX_RANK:
            raise ValueError(f"LoRA rank must be in [1, {MAX_RANK}] (got {rank})")
        self.in_features = in_features
        self.out_features = out_features
        self.rank = rank
        self.alpha = alpha
        self.scaling = alpha / rank
        self.lora_A = nn.Parameter(torch.zeros(rank, in_features))
        self.lora_B = nn.Parameter(torch.zeros(out_features, rank))
        self.dropout = nn.Dropout(p=dropout) if dropout > 0.0 else nn.Identity()
        self._packed = PackedOperands()   # bf16 operand images of A/B, re-packed when they change (not state)
        self.reset_parameters()

    def reset_parameters(self):
        # A ~ U(+-1/sqrt(in_features)) (kaiming_uniform with a=sqrt(5), fan_in = size(1)); B = 0
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B)

    @property
    def dropout_p(self) -> float:
        return float(self.dropout.p) if isinstance(self.dropout, nn.Dropout) else 0.0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return lora_linear(x, None, None, self.lora_A, self.lora_B, self.scaling, LAYOUT_PACKAGE,
                           self.dropout_p, self.training, cache=self._packed)

    def merge_weights(self) -> torch.Tensor:
        """``(lora_B @ lora_A) * scaling`` as an [out, in] fp32 matrix."""
        zero = torch.zeros(self.out_features, self.in_features, device=self.lora_A.device)
        return merge_weight(zero, self.lora_A, self.lora_B, self.scaling, LAYOUT_PACKAGE)


class LinearWithLoRA(nn.Module):
    def __init__(self, linear: nn.Linear, rank: int = 4, alpha: float = 1.0, dropout: float = 0.0):
        super().__init__()
        self.linear = linear
        for p in self.linear.parameters():
            p.requires_grad = False
        self.in_features = linear.in_features
        self.out_features = linear.out_features
        self.lora = LoRALayer(linear.in_features, linear.out_features, rank=rank, alpha=alpha,
                              dropout=dropout)
        self._wt = TransposedCopy()       # W^T of the frozen weight for the TN-form input-gradient GEMM (not state)

    # nn.MultiheadAttention and friends read these off the wrapped module
    @property
    def weight(self):
        return self.linear.weight

    @property
    def bias(self):
        return self.linear.bias

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lo = self.lora
        return lora_linear(x, self.linear.weight, self.linear.bias, lo.lora_A, lo.lora_B, lo.scaling,
                           LAYOUT_PACKAGE, lo.dropout_p, self.training, cache=lo._packed, wt_cache=self._wt)

    def merge_weights(self) -> nn.Linear:
        """A plain nn.Linear whose weight is ``W + scaling * B @ A`` (bias cloned)."""
        lo = self.lora
        merged = merge_weight(self.linear.weight, lo.lora_A, lo.lora_B, lo.scaling, LAYOUT_PACKAGE)
        out = nn.Linear(self.linear.in_features, self.linear.out_features,
                        bias=self.linear.bias is not None, device=merged.device)
        out.weight.data = merged.to(self.linear.weight.dtype)
        if self.linear.bias is not None:
            out.bias.data = self.linear.bias.data.clone()
        return out
