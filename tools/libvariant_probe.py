#!/usr/bin/env python3
"""
One library build (SAM3_LORA_AMD_LIB), one process: the adapter kernels' in-situ times and the four stand-alone calls of
bench.py's op table at BASELINE configs[1]'s sizes.  For A/B of COMPILE-TIME variants on one box: run the builds in turn, twice
(box-to-box spread on the pool is ~5 %, so only runs of one gpurun call are compared).
    for r in 1 2; do for l in base x1; do SAM3_LORA_AMD_LIB=build_exp/lib$l.so python tools/libvariant_probe.py $l; done; done
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
    rank = int(os.environ.get("PROBE_RANK", "16"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.Workload(dev, 8, rank, 8, seed=1234)
    for _ in range(2):
        w.step(recompute=False)
    torch.cuda.synchronize()
    rows = bench.insitu_kernels(w, steps=2)
    ops = bench.op_table(w, 10)
    out = {"tag": tag, "lib": os.environ.get("SAM3_LORA_AMD_LIB", "in-tree"),
           "kernels": {f"{k['kernel']}@{k['dim']}": [k["avg_us"], k["min_us"]] for k in rows},
           "ops": {o["op"]: [o["avg_us"], o["frac_of_peak"]] for o in ops}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
