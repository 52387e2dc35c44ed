#!/usr/bin/env python3
"""Same-box, same-process A/B of kernel variants selected by environment flags (read per launch by the
library).  usage: python tools/ab.py FLAG1 [FLAG2 ...]   -> baseline vs each flag=1, interleaved twice.
Box-to-box variance on the GPU pool is ~5 %, so variants are only ever compared inside one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    flags = sys.argv[1:]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.Workload(dev, 8, 16, 32, seed=1)
    w.step()
    results = {}
    order = [None] + flags
    for rep in range(2):
        for f in order:
            for g in flags:                      # "NAME" toggles 0/1, "NAME=VALUE" sets a value (unset = default)
                os.environ.pop(g.split("=")[0], None)
                if "=" not in g:
                    os.environ[g] = "0"
            if f:
                os.environ[f.split("=")[0]] = f.split("=")[1] if "=" in f else "1"
            w.step()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(3):
                w.step()
            ev1.record()
            ev1.synchronize()
            step_ms = ev0.elapsed_time(ev1) / 3
            rows = bench.insitu_kernels(w, steps=2)
            results.setdefault(f or "baseline", []).append((step_ms, rows))
    keys = sorted({(r["kernel"], r["dim"]) for v in results.values() for _, rows in v for r in rows})
    names = list(results)
    print(f"{'kernel':10s} {'dim':>5s} " + " ".join(f"{n[-22:]:>24s}" for n in names))
    for k in keys:
        line = f"{k[0]:10s} {k[1]:5d} "
        for n in names:
            vals = [r["avg_us"] for _, rows in results[n] for r in rows if (r["kernel"], r["dim"]) == k]
            line += " ".join(f"{v:11.2f}" for v in vals) + " "
        print(line)
    print(f"{'step_ms':16s} " + " ".join(" ".join(f"{s:11.3f}" for s, _ in results[n]) for n in names))


if __name__ == "__main__":
    main()
