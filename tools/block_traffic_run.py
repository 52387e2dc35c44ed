#!/usr/bin/env python3
"""The command profiled for the block-level HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, then
`tools/rocpd_summary.py block_traffic`): exactly STEPS x BLOCKS forward + backward passes of one ViT block's two adapters
(fc1 1024 -> 4736, fc2 4736 -> 1024, M = 8 x 5184, r = 16) through the plain C-ABI entry points, operands pre-packed, no recompute."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

STEPS, BLOCKS = 3, 8
if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.Workload(dev, 8, 16, BLOCKS, seed=1234)
    for _ in range(STEPS):
        w.step(recompute=False)
    torch.cuda.synchronize()
    print(STEPS * BLOCKS)
