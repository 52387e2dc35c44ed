#!/usr/bin/env python3
"""Same-box A/B of the whole-trunk training step: augmented-GEMM mode on/off (SAM3_LORA_FUSED)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for rep in range(2):
    for fused in ("1", "0"):
        os.environ["SAM3_LORA_FUSED"] = fused
        r = bench.trunk_step_bench(dev, 8, 16, 3, 1)
        print(f"fused={fused}: {r['images_per_s']} img/s  {r['ms_per_step']} ms/step  peak {r['peak_mem_gb']} GB")
