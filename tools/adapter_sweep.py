#!/usr/bin/env python3
"""
A/B of the adapter kernels' variants inside ONE process on the GPU box (interleaved rounds, bench.py's op table and in-situ
kernel timer): default (hi + lo operands, reduction riding on k_t2) against the knobs that switch each change off.
Usage: python tools/adapter_sweep.py [--batch 8] [--rank 16] > gpurun_out/adapter_sweep.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench

VARIANTS = {
    "default (hi+lo, riding reduce)": {},
    "single_round": {"SAM3_LORA_SINGLE_ROUND": "1"},
    "no_ride": {"SAM3_LORA_NO_RIDE": "1"},
    "round-2 equivalent (single_round + no_ride)": {"SAM3_LORA_SINGLE_ROUND": "1", "SAM3_LORA_NO_RIDE": "1"},
}
KNOBS = ("SAM3_LORA_T3_COOP", "SAM3_LORA_SINGLE_ROUND", "SAM3_LORA_NO_RIDE", "SAM3_LORA_T3E_WGS", "SAM3_LORA_T3_WGS", "SAM3_LORA_T2_TPW", "SAM3_LORA_XCD_ORDER",
         "SAM3_LORA_BWD_V2", "SAM3_LORA_T3W_WGS", "SAM3_LORA_TWO_PASS_GY")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--extra", action="append", default=[], help="NAME=K1=V1,K2=V2 additional variant")
    ap.add_argument("--only-extra", action="store_true", help="default + the --extra variants only")
    args = ap.parse_args()
    if args.only_extra:
        for k in list(VARIANTS)[1:]:
            del VARIANTS[k]
    for e in args.extra:
        name, _, kv = e.partition("=")
        VARIANTS[name] = dict(p.split("=") for p in kv.split(",") if p)
    from sam3_lora_amd import _ffi
    lib = _ffi.load()
    dev = torch.device("cuda", 0)
    out = {}
    for rnd in range(args.rounds):
        for name, env in VARIANTS.items():
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            lib.sam3_lora_debug_reload_knobs()
            w = bench.Workload(dev, args.batch, args.rank, 8, seed=1234)       # 8 blocks: enough launches per kernel
            for _ in range(2):
                w.step(recompute=False)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                w.step(recompute=False)
            b.record()
            b.synchronize()
            per_block_us = a.elapsed_time(b) * 1e3 / 3 / 8
            rows = bench.insitu_kernels(w, steps=1)
            ops = bench.op_table(w, 10)
            r = out.setdefault(name, {"env": env, "rounds": []})
            r["rounds"].append({"fwd+bwd_per_block_us_no_recompute": round(per_block_us, 1),
                                "ops": {o["op"]: [o["avg_us"], o["frac_of_peak"]] for o in ops},
                                "kernels": {f"{k['kernel']}@{k['dim']}": [k["avg_us"], round(k["GBps"] / 8000, 3) if k["GBps"] else None] for k in rows}})
            del w
            torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
