#!/usr/bin/env python3
"""k_t2 at large M (batch 16 / 32): tiles per workgroup sweep."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for b in (8, 16, 32):
    w = bench.Workload(dev, b, 16, 6, seed=1)
    for tpw in ("", "4", "8", "24", "48", "96"):
        os.environ.pop("SAM3_LORA_T2_TPW", None) if tpw == "" else os.environ.__setitem__("SAM3_LORA_T2_TPW", tpw)
        w.step(); rows = bench.insitu_kernels(w, steps=2)
        t2 = {r["dim"]: (r["avg_us"], r["GBps"]) for r in rows if r["kernel"] == "k_t2"}
        print(f"batch {b} tpw {tpw or 'default':8s} k_t2@4736 {t2.get(4736)}  @1024 {t2.get(1024)}", flush=True)
    del w; torch.cuda.empty_cache()
