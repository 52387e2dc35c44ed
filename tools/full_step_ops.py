#!/usr/bin/env python3
"""torch.profiler view of the whole training step: top operators by GPU time, grouped by input shapes (which PyTorch
op, on which tensor, costs what).  Usage (GPU box): python tools/full_step_ops.py [--fp8]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--rows", type=int, default=45)
    ap.add_argument("--ops", type=int, default=140)
    ap.add_argument("--host", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if a.fp8:
        from sam3_lora_amd.fp8 import enable_fp8_frozen
        enable_fp8_frozen(True)
    full = bench.FullStep(dev, 8, 16, 1, 0)
    for _ in range(3):
        full.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(2):
            full.step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    print(ka.table(sort_by="self_cuda_time_total", row_limit=a.rows, max_name_column_width=48,
                   max_shapes_column_width=70))
    print("# operators by self HOST time per step (ms): where the host is slow regardless of the device")
    host = sorted(ka, key=lambda e: -e.self_cpu_time_total)
    for e in host[:a.host]:
        print("%9.3f  %5d  %9.1f us/call  %-44s %s" % (e.self_cpu_time_total / 2e3, e.count // 2,
                                                      e.self_cpu_time_total / max(e.count, 1), e.key[:44],
                                                      str(e.input_shapes)[:110]))
    # every host-side operator (kernels themselves excluded) with its shapes: who launches the elementwise work
    print("# operators by self GPU time per step (ms), 2 steps profiled")
    ops = [e for e in ka if (e.key.startswith("aten::") or "Backward" in e.key or e.key.startswith("_"))
           and e.self_device_time_total > 100.0]
    ops.sort(key=lambda e: -e.self_device_time_total)
    for e in ops[:a.ops]:
        print("%9.3f  %5d  %-44s %s" % (e.self_device_time_total / 2e3, e.count // 2, e.key[:44],
                                        str(e.input_shapes)[:150]))


if __name__ == "__main__":
    main()
