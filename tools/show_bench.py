#!/usr/bin/env python3
"""Pretty-print the JSON line of a bench.py log (diagnostic helper)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "n_gpus", d["n_gpus"])
if "roofline" in d:
    print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_us", "traffic")})
if "step" in d.get("roofline", {}):
    print("whole step vs roofline", d["roofline"]["step"])
    if "in_sequence" in d["roofline"]:
        print("block in sequence", d["roofline"]["in_sequence"])
for k in d.get("kernels", []):
    print(f"  {k['kernel']:9s} dim={k['dim']:5d} n={k['launches']:4d} avg={k['avg_us']:8.2f} med={k['median_us']:8.2f} min={k['min_us']:8.2f} tot={k['total_us']:9.1f} {(k['GBps'] or 0):7.1f} GB/s")
for k in d.get("ops", []):
    print(f"  {k['op']:34s} {k['avg_us']:8.2f} us {k['GBps']:7.1f} GB/s frac {k['frac_of_peak']}")
if "trunk_step" in d:
    t = d["trunk_step"]
    print("trunk_step", t["images_per_s"], "img/s", t["ms_per_step"], "ms/step, peak", t["peak_mem_gb"], "GB, finite", t["loss_finite"])
if "trunk_step_no_checkpoint" in d:
    t = d["trunk_step_no_checkpoint"]
    print("trunk_step (no activation checkpointing)", t["images_per_s"], "img/s", t["ms_per_step"], "ms/step, peak", t["peak_mem_gb"], "GB")
if "no_recompute" in d:
    print("adapter path without recompute", d["no_recompute"]["value"], "img/s", d["no_recompute"]["ms_per_step"], "ms/step")
if "torch_unfused_block" in d:
    print("torch unfused block", d["torch_unfused_block"])
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
