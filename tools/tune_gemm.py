#!/usr/bin/env python3
"""Does PyTorch's TunableOp find faster hipBLASLt/rocBLAS solutions for the trunk's frozen GEMMs?
Times the whole-trunk step untuned, tunes (results -> gpurun_out/tunableop_results.csv), times again."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = bench.trunk_step_bench(dev, 8, 16, 3, 1)
print(f"untuned : {r['images_per_s']} img/s  {r['ms_per_step']} ms/step", flush=True)
import torch.cuda.tunable as tn
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tn.set_filename(os.path.join(ROOT, "gpurun_out", "tunableop_results.csv"))
tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(30)
tn.set_max_tuning_iterations(20)
r = bench.trunk_step_bench(dev, 8, 16, 2, 1)
print(f"tuning  : {r['images_per_s']} img/s  {r['ms_per_step']} ms/step", flush=True)
tn.tuning_enable(False)
tn.write_file()
r = bench.trunk_step_bench(dev, 8, 16, 3, 1)
print(f"tuned   : {r['images_per_s']} img/s  {r['ms_per_step']} ms/step", flush=True)
print(open(os.path.join(ROOT, "gpurun_out", "tunableop_results.csv")).read()[:3000])
