"""
TEST / BASELINE INFRASTRUCTURE -- not part of the product path.

The LoRA adapter written the way the reference writes it, as plain torch modules for the CPU:
``lora_layers.py:49-55`` (``((dropout(x) @ lora_A) @ lora_B) * (alpha / rank)``, A [in, r] kaiming-uniform(a=sqrt(5)),
B [r, out] zeros) and ``lora_layers.py:87-91`` (``original_layer(x) + lora(x)``), with the root injector's name rule
(basename in targets, never ``out_proj``).  Used only by ``bench.py``'s ``cpu_baseline`` leg to re-enact the
reference's CPU training step around this library's PyTorch host model; pinned through the same golden vectors as
``oracle/lora_oracle.py`` (tests/test_oracle_golden.py::test_torch_cpu_form_matches_golden).  The product's adapters
(``sam3_lora_amd.lora_layers``) never import this file and have no CPU path.
"""
import math

import torch
import torch.nn as nn


class ReferenceFormLoRALinear(nn.Module):
    def __init__(self, original_layer: nn.Linear, rank: int, alpha: float, dropout: float = 0.0):
        super().__init__()
        self.original_layer = original_layer
        a = torch.empty(original_layer.in_features, rank)
        nn.init.kaiming_uniform_(a, a=math.sqrt(5))
        self.lora_A = nn.Parameter(a)
        self.lora_B = nn.Parameter(torch.zeros(rank, original_layer.out_features))
        self.scaling = alpha / rank
        self.dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()

    def forward(self, x):
        return self.original_layer(x) + ((self.dropout(x) @ self.lora_A) @ self.lora_B) * self.scaling


def apply_reference_form_lora(model: nn.Module, rank: int, alpha: float, targets=("fc1", "fc2"), dropout: float = 0.0,
                              only_under: str = "") -> int:
    """Freeze everything, then wrap every nn.Linear whose basename is in ``targets`` (and whose qualified name contains
    ``only_under``).  Returns the number of wrapped modules."""
    for p in model.parameters():
        p.requires_grad = False
    todo = [(n, m) for n, m in model.named_modules()
            if isinstance(m, nn.Linear) and n.split(".")[-1] in targets and "out_proj" not in n and only_under in n]
    for name, lin in todo:
        parent = model
        *path, leaf = name.split(".")
        for part in path:
            parent = getattr(parent, part)
        setattr(parent, leaf, ReferenceFormLoRALinear(lin, rank, alpha, dropout))
    return len(todo)
