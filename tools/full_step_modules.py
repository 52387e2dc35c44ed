#!/usr/bin/env python3
"""Where the whole training step spends its GPU time, by module: forward and backward HIP-event brackets on the main
sub-modules of the SAM3 image model (hooks), plus matching / loss / optimizer phases.  Usage (GPU box):
    python tools/full_step_modules.py [--steps 3] [--batch 8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rank", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    full = bench.FullStep(dev, args.batch, args.rank, 1, 0)
    m = full.model
    named = {
        "trunk": m.backbone.vision_backbone.trunk,
        "neck.convs[0] (x4 -> 288^2)": m.backbone.vision_backbone.convs[0],
        "neck.convs[1] (x2 -> 144^2)": m.backbone.vision_backbone.convs[1],
        "neck.convs[2] (x1 -> 72^2)": m.backbone.vision_backbone.convs[2],
        "neck.convs[3] (x0.5, dropped by scalp)": m.backbone.vision_backbone.convs[3],
        "text tower": m.backbone.language_backbone,
        "geometry encoder": m.geometry_encoder,
        "fusion encoder": m.transformer.encoder,
        "decoder": m.transformer.decoder,
        "scoring head": m.dot_prod_scoring,
        "mask head": m.segmentation_head,
        "mask head / pixel decoder": m.segmentation_head.pixel_decoder,
        "mask head / mask predictor": m.segmentation_head.mask_predictor,
    }
    rec = []

    def ev(tag):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        rec.append((tag, e))

    for name, mod in named.items():
        mod.register_forward_pre_hook(lambda mod_, inp, n=name: ev((n, "fwd", 0)))
        mod.register_forward_hook(lambda mod_, inp, out, n=name: ev((n, "fwd", 1)))
        mod.register_full_backward_pre_hook(lambda mod_, g, n=name: ev((n, "bwd", 0)))
        mod.register_full_backward_hook(lambda mod_, gi, go, n=name: ev((n, "bwd", 1)))
    for _ in range(2):
        full.step()
    torch.cuda.synchronize()
    totals = {}
    wall = []
    for _ in range(args.steps):
        rec.clear()
        marks = []
        full.step(timers=marks)
        torch.cuda.synchronize()
        wall.append({n: (t - marks[i][1]) * 1e3 for i, (n, t) in enumerate(marks[1:])})
        open_ = {}
        for (n, ph, end), e in rec:
            if not end:
                open_[(n, ph)] = e
            elif (n, ph) in open_:
                totals.setdefault((n, ph), []).append(open_.pop((n, ph)).elapsed_time(e))
    print(f"{'module':44s} {'fwd ms':>9s} {'bwd ms':>9s}")
    for n in named:
        f = sum(totals.get((n, "fwd"), [0])) / args.steps
        b = sum(totals.get((n, "bwd"), [0])) / args.steps
        print(f"{n:44s} {f:9.2f} {b:9.2f}")
    print("\nphases (synchronised wall clock, ms):")
    for k in wall[0]:
        print(f"  {k:28s} {sum(w[k] for w in wall) / len(wall):8.2f}")


if __name__ == "__main__":
    main()
