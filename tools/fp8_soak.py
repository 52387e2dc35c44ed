#!/usr/bin/env python3
"""
Soak of the fp8 frozen-W mode at the reference's literal adapter configuration (r = 32, alpha = 64, dropout 0.1: BASELINE
configs[4]'s mode on configs/full_lora_config.yaml's adapters) -- VERDICT r5 item 6: round 4 saw ONE non-finite run in this
sequence (bf16 steps, then the switch to fp8) that 13 later runs did not reproduce.  Replays the sequence RUNS times in one
process, each time from fresh delayed-scaling state, with two injected disturbances:
  * a 3x drop of the loss (all incoming gradients shrink 3x from one step to the next: the e5m2 roles' delayed scales lag),
  * an all-zero tensor in one role (the input of a trunk block's attention projection zeroed for one step: amax = 0 observed),
and checks on the device, one host sync per run: loss and A/B gradients finite in every step, every quantiser scale finite and
positive, no amax slot non-finite.  Writes gpurun_out/r06_fp8_soak.json.

    python tools/fp8_soak.py [--runs 30] [--bf16-steps 2] [--fp8-steps 6]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from sam3_lora_amd import fp8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=30)
    ap.add_argument("--bf16-steps", type=int, default=2)
    ap.add_argument("--fp8-steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--bf16-only", action="store_true", help="control: the same sequence and disturbances with the fp8 mode never switched on")
    ap.add_argument("--trace-gemms", action="store_true", help="|in|, |out|, scale of every fp8 GEMM call, kept on the device per step; printed around the first odd one of a non-finite run")
    ap.add_argument("--rank", type=int, default=32)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--poison", action="store_true", help="fill the allocator's free memory with NaN patterns first (finds reads of unwritten memory)")
    ap.add_argument("--debug", action="store_true", help="per step: the first module with a non-finite output and its roles' state")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    full = bench.FullStep(dev, args.batch, args.rank, 1, 0, dropout=args.dropout)
    trunk = full.model.backbone.vision_backbone.trunk
    proj = trunk.blocks[5].attn.proj
    state = {"scale": 1.0, "zero": False}
    full.wrapper.register_forward_hook(lambda m, i, o: ({**o, "core_loss": o["core_loss"] * state["scale"]} if state["scale"] != 1.0 else None))
    proj.register_forward_pre_hook(lambda m, a: ((torch.zeros_like(a[0]),) + tuple(a[1:])) if state["zero"] else None)
    gemm_log = []           # --trace-gemms: (label, |in| max, |out| max, scale) of every fp8 GEMM of the current step, device tensors
    if args.trace_gemms:
        wnames = {id(p_): n for n, p_ in full.model.named_parameters()}

        def wrap(name):
            orig = getattr(fp8, name)

            def f(*a, **k):
                out = orig(*a, **k)
                w = a[2] if name.endswith("_q") else a[1]
                src = a[0]
                st = fp8.state_for(w)
                sc = a[1] if name.endswith("_q") else (st.qg.scale if "dx" in name else st.qx.scale)
                gemm_log.append((f"{name} {wnames.get(id(w), '?')}", src.detach().float().abs().max(), out.detach().float().abs().max(),
                                 sc.detach().reshape(()).clone()))
                return out
            setattr(fp8, name, f)
        for nm in ("fp8_linear", "fp8_dx", "fp8_linear_q", "fp8_dx_q"):
            wrap(nm)
    runs, worst = [], {"scale_min": float("inf"), "scale_max": 0.0}
    params0 = [p.detach().clone() for p in full.params]
    if args.poison:
        # Every byte the caching allocator will hand out from now on reads as NaN (0xFF..: NaN in fp32 / bf16 / e4m3 / e5m2): a kernel that
        # reads a position nobody wrote (torch.empty workspaces, images, padding rows) now meets NaN instead of the zeros of a fresh box.
        # The one non-finite soak run of round 6 came right after the GPU test suite had run on the same box (its NaN / Inf cases
        # left in freed memory); fresh boxes never reproduced it.
        free, _ = torch.cuda.mem_get_info(dev)
        chunks = [torch.empty(2 << 30, dtype=torch.uint8, device=dev).fill_(0xFF) for _ in range(int(free * 0.85) >> 31)]
        torch.cuda.synchronize()
        print(f"poisoned {len(chunks) * 2} GiB of device memory with 0xFF", flush=True)
        del chunks
    for run in range(args.runs):
        fp8.enable_fp8_frozen(False)                    # drops every role's delayed-scaling state
        flags, detail, step_logs = [], [], []
        gemm_log.clear()
        for _ in range(args.bf16_steps):
            full.step()
        fp8.enable_fp8_frozen(not args.bf16_only)
        for step in range(args.fp8_steps):
            # disturbances from the third fp8 step on (the first two let every role see a predecessor)
            state["scale"] = 1.0 / 3.0 if step in (2, 3) else 1.0
            state["zero"] = step == 4
            events, hooks = [], []
            if args.debug:      # first module whose output is non-finite (flags stay on the device until the step is over)
                def note(label, out):
                    t = out[0] if isinstance(out, (tuple, list)) else out
                    if isinstance(t, torch.Tensor) and t.is_floating_point():
                        events.append((label, (~torch.isfinite(t.detach().float())).any()))
                for n, m in full.model.named_modules():
                    hooks.append(m.register_forward_hook(lambda mod, inp, out, n=n: note(n, out)))
            try:
                loss = full.step()
            except Exception as e:      # (a non-finite matching cost raises inside scipy's assignment solver)
                print(f"run {run} fp8 step {step}: {type(e).__name__}: {str(e)[:100]}", flush=True)
                loss = torch.full((), float("nan"), device=dev)
                if not args.debug:
                    torch.cuda.synchronize()
                    full.opt.zero_grad(set_to_none=True) if hasattr(full, "opt") else None
                    flags.append(torch.ones((), dtype=torch.bool, device=dev))
                    break
            for h in hooks:
                h.remove()
            if args.debug:
                torch.cuda.synchronize()
                first = next((lbl for lbl, f in events if bool(f)), None)
                if first is not None:
                    print(f"run {run} fp8 step {step}: first non-finite module output: {first}", flush=True)
                    names = {id(p_): n for n, p_ in full.model.named_parameters()}
                    for key, (ref, st) in list(fp8._WEIGHTS.items()):
                        wn = names.get(id(ref()), "?")
                        if first.rsplit(".", 2)[0] not in wn:
                            continue
                        for role, q in (("x", st.qx), ("g", st.qg)):
                            if q.amax is not None:
                                a = q.amax.float().cpu()
                                print(f"   {wn} {role}: scale {float(q.scale):.4g} k {q.k} slots-max {[float(a[i][::32].max()) for i in (0, 1)]} "
                                      f"eff {[float(a[i][1]) for i in (0, 1)]}", flush=True)
                    flags.append(torch.ones((), dtype=torch.bool, device=dev))
                    break       # this run is recorded as non-finite; the next one starts from fresh state
            if args.trace_gemms:     # keep the log of a step only until the next one is known to be finite: one cheap device-side test
                worst_out = torch.stack([o for _, _, o, _ in gemm_log]).max() if gemm_log else torch.zeros((), device=dev)
                step_logs.append((step, list(gemm_log), worst_out))
                gemm_log.clear()
            per = torch.stack([~torch.isfinite(loss.detach().float()).reshape(())] +
                              [((~torch.isfinite(p.grad)).any() if p.grad is not None else torch.zeros((), dtype=torch.bool, device=dev))
                               for p in full.params])
            detail.append((step, per))
            flags.append(per.any())
        state["scale"], state["zero"] = 1.0, False
        scales, amax_bad, roles = [], 0, 0
        for key, (ref, st) in list(fp8._WEIGHTS.items()):
            for q in (st.qx, st.qg):
                if q.amax is None:
                    continue
                roles += 1
                scales.append(q.scale.reshape(-1))
                amax_bad += int((~torch.isfinite(q.amax)).any())
        sc = torch.cat(scales) if scales else torch.ones(1, device=dev)
        torch.cuda.synchronize()
        for step_, per in detail:       # the FIRST non-finite step of the run: the loss, and whose A / B gradients
            if bool(per.any()):
                pn = {id(p_): n for n, p_ in full.model.named_parameters()}
                who = [pn.get(id(p_), "?") for p_, f in zip(full.params, per[1:].tolist()) if f]
                print(f"run {run}: first non-finite fp8 step {step_}: loss non-finite {bool(per[0])}, {len(who)} of {len(full.params)} "
                      f"A/B gradients non-finite: {who[:16]}", flush=True)
                break
        if args.trace_gemms and any(bool(f) for f in flags):
            for step_, entries, worst_out in step_logs:
                vals = [(lbl, float(i), float(o), float(sc_)) for lbl, i, o, sc_ in entries]
                odd = [k for k, v in enumerate(vals) if not (v[2] < 1e6) or not (v[1] < 1e6)]
                if odd:
                    k0 = max(0, odd[0] - 6)
                    print(f"run {run} fp8 step {step_}: first fp8 GEMM with |in| or |out| >= 1e6 / non-finite is call {odd[0]} of {len(vals)}; calls {k0}..{odd[0] + 3}:", flush=True)
                    for k in range(k0, min(len(vals), odd[0] + 4)):
                        print(f"    [{k}] {vals[k][0]}: |in| {vals[k][1]:.4g} |out| {vals[k][2]:.4g} scale {vals[k][3]:.4g}", flush=True)
                    break
        params_finite = all(bool(torch.isfinite(p_).all()) for p_ in full.params)
        skipped = float(full._guard.skipped) if getattr(full, "_guard", None) is not None else 0.0
        if not params_finite:     # (only without the trainer's guard: the optimizer has stepped on non-finite gradients) restore the adapters
            with torch.no_grad():
                for p_, keep in zip(full.params, params0):
                    p_.copy_(keep)
            from sam3_lora_amd.trainer import make_adamw
            full.opt = make_adamw(full.params, lr=5e-5, weight_decay=0.01)     # (its moments are non-finite too)
        rec = {"run": run, "steps_nonfinite": [i for i, f in enumerate(flags) if bool(f)], "roles": roles,
               "scale_min": float(sc.min()), "scale_max": float(sc.max()), "scales_nonfinite": int((~torch.isfinite(sc)).sum()),
               "scales_nonpositive": int((sc <= 0).sum()), "amax_buffers_nonfinite": amax_bad, "last_loss": float(loss),
               "params_finite": params_finite, "optimizer_steps_skipped_so_far": skipped}
        worst["scale_min"] = min(worst["scale_min"], rec["scale_min"])
        worst["scale_max"] = max(worst["scale_max"], rec["scale_max"])
        runs.append(rec)
        print(json.dumps(rec), flush=True)
    fp8.enable_fp8_frozen(False)
    clean = [r for r in runs if not r["steps_nonfinite"] and not r["scales_nonfinite"] and not r["scales_nonpositive"] and not r["amax_buffers_nonfinite"]]
    # with the trainer's guard (round 6) a non-finite step is SKIPPED: a run survives when its parameters stay finite and its last loss is finite
    survived = [r for r in runs if r["params_finite"] and r["last_loss"] == r["last_loss"] and abs(r["last_loss"]) != float("inf")]
    print("runs with a non-finite step: %d; runs that survived (finite parameters and last loss): %d / %d" % (len(runs) - len(clean), len(survived), len(runs)))
    out = {"what": __doc__.strip().split("\n\n")[0], "runs": len(runs), "clean_runs": len(clean), "bf16_steps": args.bf16_steps,
           "fp8_steps": args.fp8_steps, "disturbances": {"loss_x_one_third_in_fp8_steps": [2, 3], "all_zero_proj_input_in_fp8_step": 4},
           "rank": args.rank, "dropout": args.dropout, "batch": args.batch, "scale_range_over_all_roles": worst,
           "runs_survived_with_finite_parameters": len(survived), "records": runs}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_fp8_soak.json"), "w"), indent=1)
    print("clean runs: %d / %d" % (len(clean), len(runs)))
    sys.exit(0 if len(survived) == len(runs) else 1)


if __name__ == "__main__":
    main()
