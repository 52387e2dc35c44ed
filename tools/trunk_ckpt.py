#!/usr/bin/env python3
"""Whole-trunk step with and without per-block activation checkpointing (288 GB of HBM make the recompute optional)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for ck in (True, False, True, False):
    torch.cuda.reset_peak_memory_stats(dev)
    r = bench.trunk_step_bench(dev, 8, 16, 3, 1, checkpoint=ck)
    print(f"checkpoint={ck}: {r['images_per_s']} img/s  {r['ms_per_step']} ms/step  peak {r['peak_mem_gb']} GB", flush=True)
for b in (16, 32):
    torch.cuda.reset_peak_memory_stats(dev)
    r = bench.trunk_step_bench(dev, b, 16, 2, 1, checkpoint=False)
    print(f"batch {b} checkpoint=False: {r['images_per_s']} img/s  {r['ms_per_step']} ms/step  peak {r['peak_mem_gb']} GB", flush=True)
