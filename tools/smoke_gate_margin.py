#!/usr/bin/env python3
"""
smoke()'s whole-step check compares A/B gradients of the same fp32 step across implementations (HIP adapters on the GPU,
reference-form adapters on the CPU and on the GPU).  The model has ReLU gates (FFNs of the fusion encoder / decoder / geometry
encoder, the MLP heads, GroupNorm + ReLU of the pixel decoder): an element whose pre-activation lies within the forward's
rounding noise (~1e-6) of zero takes a different gate on a different implementation and moves the adapters' gradients by 1e-3 ..
1e-2 at an identical loss (tools/backward_gpu_vs_cpu.py found exactly one such element, |pre-activation| = 1.1e-7, behind the
"4e-3 wobble" of rounds 2-3).  This script evaluates the CPU leg for a range of model seeds and prints the smallest
|pre-activation| over all ReLU gates, so that smoke() can use a seed with a margin of >= 1e-5.  Runs on the CPU.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F


def gate_margin(seed: int):
    import __graft_entry__ as G
    seen = []
    orig = F.relu

    def relu(x, inplace=False):
        if x.numel():
            seen.append(float(x.detach().abs().min()))
        return orig(x, inplace=inplace)
    F.relu = relu
    try:
        loss, _ = G._whole_step("reference", "cpu", seed)
    finally:
        F.relu = orig
    return min(seen), len(seen), loss


if __name__ == "__main__":
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        m, n, loss = gate_margin(seed)
        print(f"seed {seed}: smallest |pre-activation| over {n} ReLU calls {m:.3e}   loss {loss:.6f}", flush=True)
