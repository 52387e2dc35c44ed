#!/usr/bin/env python3
"""Does Python's cyclic garbage collector stall the training step?  Logs every collection (generation, duration) during
whole steps, then times the step (a) as is, (b) after gc.freeze(), (c) with automatic collection off.
Usage (GPU box): python tools/gc_probe.py"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def timed(full, n=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        full.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    full = bench.FullStep(dev, 8, 16, 1, 0)
    for _ in range(3):
        full.step()
    torch.cuda.synchronize()
    log, t_start = [], [0.0]

    def cb(phase, info):
        if phase == "start":
            t_start[0] = time.perf_counter()
        else:
            log.append((info["generation"], (time.perf_counter() - t_start[0]) * 1e3, info["collected"]))
    gc.callbacks.append(cb)
    print("thresholds", gc.get_threshold(), "objects tracked", len(gc.get_objects()))
    for k in range(4):
        log.clear()
        t0 = time.perf_counter()
        full.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        by_gen = {}
        for g, ms, _ in log:
            a = by_gen.setdefault(g, [0, 0.0, 0.0])
            a[0] += 1; a[1] += ms; a[2] = max(a[2], ms)
        print(f"step {k}: {dt:.1f} ms; collections " + ", ".join(f"gen{g}: {c} x, {ms:.1f} ms total, longest {mx:.1f} ms"
                                                                 for g, (c, ms, mx) in sorted(by_gen.items())))
    gc.callbacks.remove(cb)
    print(f"(a) as is:                 {timed(full):.1f} ms / step")
    gc.collect()
    gc.freeze()
    print(f"(b) after gc.freeze():     {timed(full):.1f} ms / step   (frozen objects: {gc.get_freeze_count()})")
    gc.disable()
    print(f"(c) automatic gc disabled: {timed(full):.1f} ms / step")
    gc.enable()


if __name__ == "__main__":
    main()
