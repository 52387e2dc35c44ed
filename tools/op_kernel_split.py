#!/usr/bin/env python3
"""The four stand-alone adapter calls of bench.py's op table, each repeated back to back as the table times them, with the library's
in-situ timer on: per-kernel durations INSIDE that repetition (against the same kernels inside the step sequence, bench.insitu_kernels).
    python tools/op_kernel_split.py
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from sam3_lora_amd import _ffi
from sam3_lora_amd.functional import lora_bwd_, lora_fwd_, pack_operands


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.Workload(dev, 8, 16, 2, seed=1234)
    w.step(recompute=False)
    lib = _ffi.load()
    M, r, s = w.M, w.rank, w.scaling
    x1, h, y2, g2, gh, g1 = w.x1[0], w.h[0], w.y2[0], w.g2[0], w.gh[0], w.g1[0]
    A1, B1, A2, B2 = w.A1[0], w.B1[0], w.A2[0], w.B2[0]
    gA1, gB1, gA2, gB2 = (torch.zeros_like(p) for p in (A1, B1, A2, B2))
    p1, p2 = pack_operands(A1, B1, 0, dtype=x1.dtype), pack_operands(A2, B2, 0, dtype=x1.dtype)
    t1 = lora_fwd_(x1, A1, B1, h, s, 0, save_t=True, packed=p1)
    t2 = lora_fwd_(h, A2, B2, y2, s, 0, save_t=True, packed=p2)
    ops = {"fwd fc1": lambda: lora_fwd_(x1, A1, B1, h, s, 0, packed=p1), "fwd fc2": lambda: lora_fwd_(h, A2, B2, y2, s, 0, packed=p2),
           "bwd fc1": lambda: lora_bwd_(gh, x1, t1, A1, B1, g1, gA1, gB1, s, 0, accumulate=True, packed=p1),
           "bwd fc2": lambda: lora_bwd_(g2, h, t2, A2, B2, gh, gA2, gB2, s, 0, accumulate=True, packed=p2)}
    names = {_ffi.STAGE_T1: "k_t1", _ffi.STAGE_T2: "k_t2", _ffi.STAGE_T3_GB: "k_t3(gB)", _ffi.STAGE_T3_GA: "k_t3(gA)", _ffi.STAGE_REDUCE: "k_reduce",
             64: "k_gt_reduce", _ffi.STAGE_T3W: "k_t3w", _ffi.STAGE_PACK: "k_pack"}
    # the same calls alternating between the workload's two activation sets (as consecutive blocks of the step do)
    alt = {"n": 0}
    t1b = lora_fwd_(w.x1[1], A1, B1, w.h[1], s, 0, save_t=True, packed=p1)
    t2b = lora_fwd_(w.h[1], A2, B2, w.y2[1], s, 0, save_t=True, packed=p2)

    def alternating(kind):
        def fn():
            k = alt["n"] & 1
            alt["n"] += 1
            if kind == "fwd fc1":
                lora_fwd_(w.x1[k], A1, B1, w.h[k], s, 0, packed=p1)
            elif kind == "fwd fc2":
                lora_fwd_(w.h[k], A2, B2, w.y2[k], s, 0, packed=p2)
            elif kind == "bwd fc1":
                lora_bwd_(w.gh[k], w.x1[k], (t1, t1b)[k], A1, B1, w.g1[k], gA1, gB1, s, 0, accumulate=True, packed=p1)
            else:
                lora_bwd_(w.g2[k], w.h[k], (t2, t2b)[k], A2, B2, w.gh[k], gA2, gB2, s, 0, accumulate=True, packed=p2)
        return fn
    for kind in list(ops):
        ops[kind + " (two buffer sets alternating)"] = alternating(kind)
    out = {}
    for name, fn in ops.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        b.synchronize()
        whole = a.elapsed_time(b) * 100.0
        cap = 256
        _ffi.check(lib.sam3_lora_prof_start(_ffi.STAGE_ALL, cap), "prof_start")
        for _ in range(10):
            fn()
        us, st, dm = (ctypes.c_float * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
        n = lib.sam3_lora_prof_stop(us, st, dm, cap)
        per = {}
        for i in range(n):
            per.setdefault(f"{names.get(st[i], st[i])}@{dm[i]}", []).append(us[i])
        out[name] = {"call_us_untimed_kernels": round(whole, 1), "kernels_us": {k: round(sum(v) / len(v), 1) for k, v in per.items()},
                     "kernel_sum_us": round(sum(sum(v) / len(v) for v in per.values()), 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
