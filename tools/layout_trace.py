#!/usr/bin/env python3
"""Where do the layout copies of the convolutional tail come from?  One whole training step (GPU) with (a) forward
hooks that print the memory format entering / leaving every conv / norm of neck and mask head and (b) a dispatch-mode
log of every aten copy / clone / contiguous on a 4-D tensor of >= 64 MB with the Python call site (forward) or the
autograd node (backward)."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench


def fmt(t):
    if not isinstance(t, torch.Tensor) or t.dim() != 4:
        return "-"
    if t.is_contiguous():
        return "NCHW"
    if t.is_contiguous(memory_format=torch.channels_last):
        return "NHWC"
    return "strided" + str(tuple(t.stride()))


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    full = bench.FullStep(dev, 8, 16, 1, 0)
    for _ in range(2):
        full.step()
    torch.cuda.synchronize()
    hooks = []
    for name, m in full.model.named_modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.GroupNorm, torch.nn.MaxPool2d)):
            hooks.append(m.register_forward_hook(
                lambda mod, a, out, name=name: print(f"  {name:70s} {type(mod).__name__:16s} {tuple(a[0].shape)} {fmt(a[0])} -> {fmt(out)}")))

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            a0 = args[0] if args else None
            if isinstance(a0, torch.Tensor) and a0.dim() == 4 and a0.numel() * a0.element_size() >= (48 << 20) and \
                    any(k in name for k in ("copy_", "clone", "contiguous", "_to_copy", "index", "group_norm", "add.Tensor")):
                src = args[1] if len(args) > 1 and isinstance(args[1], torch.Tensor) else None
                st = [f"{f.filename.split('/')[-1]}:{f.lineno}" for f in traceback.extract_stack() if "sam3_lora_amd" in f.filename][-3:]
                print(f"    {name:34s} {tuple(a0.shape)} {fmt(a0)}{' <- ' + fmt(src) if src is not None else ''} -> "
                      f"{fmt(out) if isinstance(out, torch.Tensor) else '-'}   {' '.join(st) or '(backward)'}")
            return out
    with Log():
        full.step()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
