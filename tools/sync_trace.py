#!/usr/bin/env python3
"""Host <-> device crossings inside one whole training step: every operator that moves data between the CPU and the GPU
or reads a device scalar on the host (each one blocks the host until the stream has drained), with its call site.
Usage (GPU box): python tools/sync_trace.py"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

import bench

SYNCING = ("_local_scalar_dense", "nonzero", "masked_select", "_unique", "unique_dim", "item")


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    full = bench.FullStep(dev, 8, 16, 1, 0)
    for _ in range(2):
        full.step()
    torch.cuda.synchronize()
    seen = collections.OrderedDict()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            ins = [t for t in tree_flatten((args, kwargs or {}))[0] if isinstance(t, torch.Tensor)]
            outs = [t for t in tree_flatten(out)[0] if isinstance(t, torch.Tensor)]
            dev_in = {t.device.type for t in ins}
            dev_out = {t.device.type for t in outs}
            cross = ("cpu" in dev_in and "cuda" in dev_out) or ("cuda" in dev_in and "cpu" in dev_out) or \
                    ("cpu" in dev_in and "cuda" in dev_in and "copy" in name)
            if cross or any(k in name for k in SYNCING):
                st = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()
                      if "/sam3_lora_amd/" in f.filename or f.filename.endswith("bench.py")][-3:]
                key = (name, " < ".join(reversed(st)) or "(autograd thread)")
                seen[key] = seen.get(key, 0) + 1
            return out
    with Log():
        full.step()
    torch.cuda.synchronize()
    print("# host <-> device crossings of one training step (operator, count, call site innermost first)")
    for (name, site), c in seen.items():
        print(f"{c:4d}  {name:38s} {site}")


if __name__ == "__main__":
    main()
