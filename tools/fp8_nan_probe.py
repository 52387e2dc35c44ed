#!/usr/bin/env python3
"""Debug aid: whole-model steps in the fp8 frozen-W mode until something non-finite appears; reports the first module whose
output is non-finite and the quantiser states (amax slots, scale) that are not finite.  GPU box: python tools/fp8_nan_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from sam3_lora_amd import fp8

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
batch = int(os.environ.get("PROBE_BATCH", "8"))
full = bench.FullStep(dev, batch, 16, 1, 0)
for _ in range(2):
    full.step()
print("bf16 loss", float(full.last_loss))
fp8.enable_fp8_frozen(True)
names = {id(p): n for n, p in full.model.named_parameters()}


def report_states():
    bad = 0
    for key, (ref, st) in list(fp8._WEIGHTS.items()):
        w = ref()
        for role, q in (("x", st.qx), ("g", st.qg)):
            if q.amax is None:
                continue
            a, s = q.amax.float().cpu(), q.scale.float().cpu()
            if not torch.isfinite(a).all() or not torch.isfinite(s).all() or float(a.max()) == 0.0:
                print("  state", names.get(id(w), "?"), role, "amax max/min", float(a.max()), float(a.min()), "scale", float(s), "k", q.k)
                bad += 1
    print("  states checked:", len(fp8._WEIGHTS), "suspicious:", bad)


if os.environ.get("PROBE_NO_PRODUCERS") == "1":
    fp8.producer_slots = lambda *a, **k: None
    print("producers disabled: every fp8 image comes from the separate quantiser")
ASYNC = os.environ.get("PROBE_SYNC", "0") != "1"       # default: record non-finite flags ON THE DEVICE (no host sync inside a step)
events = []                 # (label, device flag tensor)


def note(label, t):
    events.append((label, (~torch.isfinite(t.detach().float())).any()))


def wrap(name):
    orig = getattr(fp8, name)

    def f(*a, **k):
        w = a[2] if name.endswith("_q") else a[1]
        note(f"{name} IN  {names.get(id(w), '?')}", a[0])
        out = orig(*a, **k)
        note(f"{name} OUT {names.get(id(w), '?')}", out)
        if name.endswith("_q"):
            note(f"{name} SCALE {names.get(id(w), '?')}", a[1])
        return out
    setattr(fp8, name, f)


for nm in ("fp8_linear", "fp8_dx", "fp8_linear_q", "fp8_dx_q"):
    wrap(nm)
for step in range(8):
    events.clear()
    hooks = []
    for n, m in full.model.named_modules():
        hooks.append(m.register_forward_hook(lambda mod, inp, out, n=n: note("module " + n, (out[0] if isinstance(out, (tuple, list)) else out))
                                             if isinstance((out[0] if isinstance(out, (tuple, list)) else out), torch.Tensor)
                                             and (out[0] if isinstance(out, (tuple, list)) else out).is_floating_point() else None))
    err = None
    try:
        loss = full.step()
    except Exception as e:
        err = f"{type(e).__name__}: {str(e)[:80]}"
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    flags = [(lab, bool(f)) for lab, f in events]
    bad = [lab for lab, f in flags if f]
    print("fp8 step", step, "loss", None if err else float(loss), "error", err, "| first non-finite events:", bad[:6])
    if bad or err:
        report_states()
        break
