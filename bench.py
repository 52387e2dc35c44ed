#!/usr/bin/env python3
"""
bench.py -- SAM3-LoRA training throughput on MI355X at BASELINE.json's configuration, and the HBM roofline of the
hand-written adapter kernels inside it.

One STEP = one training step of `full_lora_config.yaml` at r=16 (BASELINE configs[1]: batch 8 per GPU of synthetic
1024^2 images -> 1008^2 model input, bf16) on the whole SAM3 image model (this library's restatement, 840.5M
parameters, random seeded init -- there is no network for the checkpoint): forward, Hungarian matching of the final and
the five auxiliary decoder outputs, Sam3LossWrapper, backward, all-reduce of the A/B gradients (N > 1), AdamW on A/B
(train_sam3_lora_native.py:887-943).  `value` = global images per second of that step; inputs are resident in HBM
before the timed region; nothing is skipped inside it.

Beside it, on the same JSON line:
  "adapter_path"  the LoRA adapter path on its own (64 adapted ViT-MLP Linears: pack + forward + checkpoint recompute +
                  backward through the C-ABI) -- the quantity the HIP kernels determine, with its CPU port;
  "roofline"      dominant adapter kernel, timed in situ with HIP events inside the library, against 8 TB/s;
                  "roofline.step" = the whole adapter path against SURVEY 8(d)'s algorithmic bytes;
  "kernels"/"ops" every kernel x shape and every C-ABI call of the adapter path;
  "mfma_bound"    the whole step against the ~140 images/s MFMA bound of SURVEY 8(d);
  "phases_ms"     forward / matching / loss / backward / exchange+optimizer of one synchronised step;
  "cpu_baseline"  the reference's CPU training step (configs[0]) re-enacted on the host cores, bounded sample;
  "distributed"   backend, RCCL version, device id of every rank;  "exchange_overlap" (N > 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BOUND_IPS = 140.0       # ~18 TFLOP per image and step at 2.5 PFLOP/s dense bf16
D_MODEL, D_HID, TOKENS, N_BLOCKS = 1024, 4736, 5184, 32


def emit_line(out):
    """The ONE JSON line, as the LAST thing on stdout: RCCL prints a version banner through C stdio when a communicator is created (also the
    one-rank communicator of the N = 1 exchange measurement), and a block-buffered C stream would otherwise be flushed at exit, behind it."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (weak scaling)")
    ap.add_argument("--blocks", type=int, default=N_BLOCKS)
    ap.add_argument("--dropout", type=float, default=0.0, help="LoRA dropout p (literal full_lora_config.yaml: 0.1)")
    ap.add_argument("--act-dtype", choices=["bf16", "f32"], default="bf16",
                    help="activation dtype (f32 = the reference's un-autocast precision; contracted as bf16 on the MFMAs)")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2,
                    help="cpu_baseline by SURVEY 8(d)'s c1 protocol: 1 warm-up + N timed B=2 CPU steps (default 2: ~5 minutes of "
                         "host time on rank 0); 0 = one bounded step")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--trunk", action="store_true", help="also time the ViT trunk alone (with / without recompute)")
    ap.add_argument("--trunk-steps", type=int, default=3)
    ap.add_argument("--adapter-only", action="store_true", help="skip the whole-model step; the line is the adapter path")
    ap.add_argument("--act-checkpoint", choices=["auto", "on", "off"], default="auto",
                    help="activation checkpointing of the whole-model step (auto: off when the activations fit in HBM)")
    ap.add_argument("--match-twice", action="store_true", help="also match inside the model's forward, as the reference does")
    ap.add_argument("--no-overlap", action="store_true", help="skip the exchange-overlap measurement (N > 1)")
    ap.add_argument("--no-data-step", action="store_true", help="skip the loader-in-the-loop measurement")
    ap.add_argument("--full-only", action="store_true", help="only the whole-model step (profiling runs)")
    ap.add_argument("--fp8-frozen", action="store_true",
                    help="headline step with fp8 frozen-W base GEMMs (BASELINE configs[4] mode; default: measured as a sub-object)")
    ap.add_argument("--no-fp8", action="store_true", help="skip the fp8 frozen-W sub-measurement")
    ap.add_argument("--no-fused-ab", action="store_true", help="skip the fused-fc1 on/off sub-measurement")
    ap.add_argument("--no-literal", action="store_true", help="skip the literal full_lora_config.yaml (r=32, dropout 0.1) sub-measurement")
    ap.add_argument("--no-fp32-layout", action="store_true", help="skip the exact-fp32-layout whole-step sub-measurement")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gates run with the timing (BASELINE.md section 3)")
    ap.add_argument("--model", choices=["sam3", "tiny"], default="sam3",
                    help="tiny: the parity fixture's widths at 112^2 (contract tests only; the line says so)")
    return ap.parse_args()


class Workload:
    """Synthetic, HBM-resident state of the adapter path of one ViT trunk."""

    def __init__(self, dev, batch, rank, blocks, seed, dropout=0.0, act_dtype=torch.bfloat16):
        from sam3_lora_amd.ddp import LoRAGradReducer
        self.dev, self.rank, self.blocks = dev, rank, blocks
        self.M = batch * TOKENS
        self.scaling = 2.0  # alpha = 2*rank in every shipped config
        self.dropout = dropout
        g = torch.Generator(device=dev).manual_seed(seed)
        M = self.M

        def act(cols):
            return torch.randn(M, cols, device=dev, generator=g, dtype=torch.float32).to(act_dtype)

        # two rotating activation sets so consecutive blocks never reuse the same lines
        self.x1 = [act(D_MODEL) for _ in range(2)]
        self.h = [act(D_HID) for _ in range(2)]
        self.y2 = [act(D_MODEL) for _ in range(2)]
        self.g2 = [act(D_MODEL) for _ in range(2)]
        self.gh = [act(D_HID) for _ in range(2)]
        self.g1 = [act(D_MODEL) for _ in range(2)]
        P = torch.nn.Parameter
        self.A1 = [P((torch.rand(D_MODEL, rank, device=dev, generator=g) - .5) * (2 / rank ** .5)) for _ in range(blocks)]
        self.B1 = [P(torch.randn(rank, D_HID, device=dev, generator=g) * 2e-3) for _ in range(blocks)]
        self.A2 = [P((torch.rand(D_HID, rank, device=dev, generator=g) - .5) * (2 / rank ** .5)) for _ in range(blocks)]
        self.B2 = [P(torch.randn(rank, D_MODEL, device=dev, generator=g) * 2e-3) for _ in range(blocks)]
        params = []
        for b in range(blocks):
            params += [self.A1[b], self.B1[b], self.A2[b], self.B2[b]]
        self.reducer = LoRAGradReducer(params, bucket_bytes=8 << 20)
        self.params = params
        self.tT1 = [None] * blocks       # t^T saved by the forward (no-recompute schedule)
        self.tT2 = [None] * blocks
        self.P1 = [None] * blocks        # operand images of (A1, B1) / (A2, B2), re-packed every step
        self.P2 = [None] * blocks

    def step(self, recompute=True):
        """recompute=True: the reference's schedule (every block re-evaluated in backward under activation
        checkpointing).  recompute=False: the schedule 288 GB of HBM allow -- t saved by the one forward."""
        from sam3_lora_amd.functional import lora_bwd_, lora_fwd_, pack_operands
        s, L, dp = self.scaling, 0, self.dropout
        prepack = os.environ.get("BENCH_PREPACK", "1") != "0"
        dbg = os.environ.get("BENCH_DEBUG_TIMING") == "1"
        marks = []

        def mark(name):
            if dbg:
                torch.cuda.synchronize()
                marks.append((name, time.perf_counter()))
        mark("start")
        self.reducer.zero_grad()
        mark("zero_grad")
        with torch.no_grad():
            if prepack:         # A/B changed at the optimizer step: pack ALL adapters once (4 launches for 64 of them);
                                # forward, checkpoint recompute and backward of the step use these images
                from sam3_lora_amd.functional import pack_operands_many
                pairs = [(self.A1[b], self.B1[b]) for b in range(self.blocks)] + [(self.A2[b], self.B2[b]) for b in range(self.blocks)]
                blobs = pack_operands_many(pairs, L, dtype=self.x1[0].dtype, outs=self.P1 + self.P2)
                self.P1, self.P2 = blobs[:self.blocks], blobs[self.blocks:]
            for b in range(self.blocks):                       # forward
                k = b & 1
                p1, p2 = (self.P1[b], self.P2[b]) if prepack else (None, None)
                t1 = lora_fwd_(self.x1[k], self.A1[b], self.B1[b], self.h[k], s, L, save_t=not recompute, drop_p=dp,
                               seed=2 * b, packed=p1)
                t2 = lora_fwd_(self.h[k], self.A2[b], self.B2[b], self.y2[k], s, L, save_t=not recompute, drop_p=dp,
                               seed=2 * b + 1, packed=p2)
                self.tT1[b], self.tT2[b] = t1, t2
            mark("pack + forward")
            for b in reversed(range(self.blocks)):             # per-block recompute, then backward
                k = b & 1
                p1, p2 = (self.P1[b], self.P2[b]) if prepack else (None, None)
                if recompute:
                    t1 = lora_fwd_(self.x1[k], self.A1[b], self.B1[b], self.h[k], s, L, save_t=True, drop_p=dp, seed=2 * b,
                                   packed=p1)
                    t2 = lora_fwd_(self.h[k], self.A2[b], self.B2[b], self.y2[k], s, L, save_t=True, drop_p=dp,
                                   seed=2 * b + 1, packed=p2)
                else:
                    t1, t2 = self.tT1[b], self.tT2[b]
                lora_bwd_(self.g2[k], self.h[k], t2, self.A2[b], self.B2[b], self.gh[k],
                          self.A2[b].grad, self.B2[b].grad, s, L, accumulate=True, drop_p=dp, seed=2 * b + 1, packed=p2)
                lora_bwd_(self.gh[k], self.x1[k], t1, self.A1[b], self.B1[b], self.g1[k],
                          self.A1[b].grad, self.B1[b].grad, s, L, accumulate=True, drop_p=dp, seed=2 * b, packed=p1)
                for p in (self.A1[b], self.B1[b], self.A2[b], self.B2[b]):
                    self.reducer.notify(p)
        mark("recompute + backward")
        self.reducer.finish()
        mark("exchange")
        if dbg:
            print("[adapter step] " + ", ".join(f"{n} {1e3 * (t - marks[i][1]):.1f} ms" for i, (n, t) in enumerate(marks[1:])),
                  file=sys.stderr, flush=True)


def time_events(fn, iters, warm=3, reps=5):
    """Average GPU time (us) of one fn() call: HIP events on the current stream bracket a batch of
    `iters` back-to-back calls (the queue stays full, so host launch latency is not in the number);
    repeated `reps` times -> (mean, median, min) of the per-call averages."""
    for _ in range(warm):
        fn()
    per = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        b.synchronize()
        per.append(a.elapsed_time(b) * 1e3 / iters)
    per.sort()
    return sum(per) / len(per), per[len(per) // 2], per[0]


def workload_label(args, n_adapted, ckpt):
    """`config.workload` of the JSON line: which BASELINE configuration this run is, and what the step contains."""
    which_config = ("BASELINE configs[4]'s adapter / precision / batch at the reference's 1008^2 -- its 1536^2 inputs do not exist "
                    "in the reference (SURVEY F5)" if (args.rank == 8 and args.batch == 16 and args.fp8_frozen)
                    else "BASELINE configs[1]" if (args.rank == 16 and args.batch == 8 and not args.fp8_frozen)
                    else "a variation of BASELINE configs[1]")
    return (("" if args.model == "sam3" else "[TINY-WIDTH CONTRACT-TEST MODEL, not the benchmark] ") +
            "full_lora_config.yaml @ r=%d alpha=%d (%s): SAM3 image model "
            "(840.5M parameters, random seeded init), batch %d/GPU of synthetic 1024^2 images -> "
            "1008^2 with 2 boxes + masks each and the prompt 'crack'; forward + Hungarian "
            "matching (final + 5 aux outputs) + Sam3LossWrapper (boxes, IA-BCE + presence, "
            "mask focal + dice, o2m twins) + backward + A/B-gradient all-reduce + AdamW; frozen "
            "tensors and activations %s, A/B fp32; LoRA on the %d ViT-MLP Linears through the "
            "HIP adapter path; activation checkpointing %s; matching %s per step%s"
            % (args.rank, 2 * args.rank, which_config, args.batch, args.act_dtype, n_adapted,
               "on (per block / layer)" if ckpt else "off (activations kept in HBM)",
               "twice (model + loop, as the reference)" if args.match_twice else "once",
               "; frozen base GEMMs in fp8 (e4m3 weights / activations, e5m2 gradients)" if args.fp8_frozen else ""))


def committed_traffic(kernel, wg_x):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*traffic.json,
    produced by tools/rocpd_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs
    of this bench; FETCH_SIZE doubled per MI355X_MICROARCH.md)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            for row in json.load(open(path)):
                if row["kernel"].startswith(kernel) and row["wg_x"] == wg_x:
                    best = dict(row, source=os.path.relpath(path, ROOT))
        except Exception:
            pass
    return best


def committed_block_traffic():
    """PMC HBM bytes of one block's forward + backward adapter calls, summed over their kernels from the newest committed pass
    (profiles/*block_traffic.json, written by tools/rocpd_summary.py --block from separate FETCH_SIZE / WRITE_SIZE runs)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*block_traffic.json"))):
        try:
            d = json.load(open(path))
            best = {"hbm_bytes": int(d["hbm_bytes_per_block"]), "source": os.path.relpath(path, ROOT)}
        except Exception:
            pass
    return best


def insitu_kernels(w, steps=2):
    """Per-kernel durations measured INSIDE the step sequence: the library brackets every launch with a
    pair of HIP events on the stream it launches on (sam3_lora_prof_start/stop), for `steps` extra steps
    after the timed region.  Grouped by (kernel, streamed dimension)."""
    import ctypes
    from sam3_lora_amd import _ffi
    lib = _ffi.load()
    cap = 16384
    _ffi.check(lib.sam3_lora_prof_start(_ffi.STAGE_ALL, cap), "prof_start")
    for _ in range(steps):
        w.step()
    us = (ctypes.c_float * cap)()
    st = (ctypes.c_int * cap)()
    dm = (ctypes.c_int * cap)()
    n = lib.sam3_lora_prof_stop(us, st, dm, cap)
    if n < 0:
        raise RuntimeError(_ffi.last_error())
    M, r = w.M, w.rank
    e = w.x1[0].element_size()
    single = os.environ.get("SAM3_LORA_SINGLE_ROUND", "0") not in ("", "0")
    RG = 16 if r <= 16 else 32                                   # rank tile(s) of one group
    hl_max = int(os.environ.get("SAM3_LORA_HL_MAX_RANK", "32") or 32)
    RP = RG * (2 if (e == 2 and not single and r <= hl_max) else 1)     # bf16: t / gt travel as hi | lo pairs (2 RG columns)
    one_pass = r <= 16 and os.environ.get("SAM3_LORA_TWO_PASS_GY", "0") in ("", "0")
    names = {_ffi.STAGE_PACK: "k_pack", _ffi.STAGE_T1: "k_t1", _ffi.STAGE_T2: "k_t2",
             _ffi.STAGE_T3_GB: "k_t3+gt" if one_pass else "k_t3", _ffi.STAGE_T3_GA: "k_t3",
             _ffi.STAGE_REDUCE: "k_reduce", 64: "k_gt_reduce", _ffi.STAGE_T3W: "k_t3w", _ffi.STAGE_XGX: "k_xgx"}

    def alg_bytes(kernel, dim):
        if kernel == "k_t2":      # read + write Y[M,N]; read T[M,RP], W2t[N,RP]
            return 2 * e * M * dim + e * M * RP + e * dim * RP
        if kernel == "k_t1":      # read X[M,K], W1[RP,K]; write T and TT
            return e * M * dim + e * RP * dim + 2 * e * M * RP
        if kernel == "k_t3":      # read X[M,N], TT[RP,M] (partials are overhead, not algorithmic)
            return e * M * dim + e * RP * M
        if kernel == "k_t3+gt":   # one pass over gy: gB partials and gt (T, TT written by k_gt_reduce)
            return e * M * dim + e * RP * M
        if kernel == "k_t3w":     # backward version 2 over gy: gB partials and gt (partials, or its images) from one read
            return e * M * dim + e * RP * M
        if kernel == "k_xgx":     # backward version 2 over x and gx: read x, read + write gx; A_c image (gt arrives as partials: overhead)
            return 3 * e * M * dim + e * dim * RP
        if kernel == "k_gt_reduce":   # pure overhead of the one-pass form: fp32 gt partials back in, T and TT out
            return -(-dim // 128) * M * 64 + 2 * e * M * RP
        if kernel == "k_reduce":  # read-modify-write fp32 gA, gB
            return 2 * 4 * r * dim
        return (4 + 2) * r * dim  # k_pack: read fp32, write bf16

    groups = {}
    for i in range(n):
        groups.setdefault((names[st[i]], dm[i]), []).append(us[i])
    rows = []
    for (k, dim), v in groups.items():
        v.sort()
        avg = sum(v) / len(v)
        nb = alg_bytes(k, dim)
        row = dict(kernel=k, dim=dim, launches=len(v), algorithmic_bytes=nb, avg_us=round(avg, 2),
                   median_us=round(v[len(v) // 2], 2), min_us=round(v[0], 2), total_us=round(sum(v), 1),
                   GBps=round(nb / avg / 1e3, 1))
        if k in ("k_gt_reduce", "k_reduce", "k_pack"):
            # their inputs are partials / masters the previous kernel just wrote: served from L2 / Infinity Cache (PMC:
            # profiles/*traffic.json, k_gt_reduce@4736 fetches 36 MB of the 104 MB it reads), so bytes / time is not an HBM rate
            row["GBps"] = None
            row["note"] = "reads cache-resident partials: not an HBM stream (see profiles/*_traffic.json for its PMC bytes)"
        rows.append(row)
    rows.sort(key=lambda r_: -r_["total_us"])
    return rows


def op_table(w, iters):
    """Whole C-ABI calls (all their kernels) against SURVEY section 8(d)'s per-unit algorithmic bytes."""
    from sam3_lora_amd.functional import lora_bwd_, lora_fwd_, pack_operands
    M, r, s, e = w.M, w.rank, w.scaling, w.x1[0].element_size()
    x1, h, y2, g2, gh, g1 = w.x1[0], w.h[0], w.y2[0], w.g2[0], w.gh[0], w.g1[0]
    A1, B1, A2, B2 = w.A1[0], w.B1[0], w.A2[0], w.B2[0]
    gA1, gB1, gA2, gB2 = (torch.zeros_like(p) for p in (A1, B1, A2, B2))
    p1, p2 = pack_operands(A1, B1, 0, dtype=x1.dtype), pack_operands(A2, B2, 0, dtype=x1.dtype)     # as the step calls them: operands pre-packed
    t1 = lora_fwd_(x1, A1, B1, h, s, 0, save_t=True, packed=p1)
    t2 = lora_fwd_(h, A2, B2, y2, s, 0, save_t=True, packed=p2)
    fwd1 = lambda: lora_fwd_(x1, A1, B1, h, s, 0, packed=p1)
    fwd2 = lambda: lora_fwd_(h, A2, B2, y2, s, 0, packed=p2)
    bwd2 = lambda: lora_bwd_(g2, h, t2, A2, B2, gh, gA2, gB2, s, 0, accumulate=True, packed=p2)
    bwd1 = lambda: lora_bwd_(gh, x1, t1, A1, B1, g1, gA1, gB1, s, 0, accumulate=True, packed=p1)
    D, H = D_MODEL, D_HID
    fwd_b = lambda i, o: e * M * (i + 2 * o) + e * r * (i + o)
    bwd_b = lambda i, o: e * M * (o + i + 2 * i) + 4 * r * (i + o) * 2
    ops = []
    for name, fn, nb in (("sam3_lora_fwd fc1 (1024->4736)", fwd1, fwd_b(D, H)),
                         ("sam3_lora_fwd fc2 (4736->1024)", fwd2, fwd_b(H, D)),
                         ("sam3_lora_bwd fc1 (1024->4736)", bwd1, bwd_b(D, H)),
                         ("sam3_lora_bwd fc2 (4736->1024)", bwd2, bwd_b(H, D))):
        avg, med, mn = time_events(fn, iters)
        ops.append(dict(op=name, algorithmic_bytes=nb, avg_us=round(avg, 2), GBps=round(nb / avg / 1e3, 1),
                        frac_of_peak=round(nb / avg / 1e3 / HBM_PEAK_GBPS, 4)))
    tot_b = sum(o["algorithmic_bytes"] for o in ops)
    tot_t = sum(o["avg_us"] for o in ops)
    ops.append(dict(op="fwd+bwd of one block (fc1+fc2)", algorithmic_bytes=tot_b, avg_us=round(tot_t, 2),
                    GBps=round(tot_b / tot_t / 1e3, 1), frac_of_peak=round(tot_b / tot_t / 1e3 / HBM_PEAK_GBPS, 4)))
    return ops


def torch_unfused_block(w, iters=5):
    """The same block (fc1 + fc2 adapters, forward + backward) written the way the reference writes it --
    ``base + ((x @ A) @ B) * s`` under torch.autograd (lora_layers.py:49-55,87-91) -- on the same GPU in bf16: the
    "before" of SURVEY section 8(d).  Returns microseconds per block (forward + backward, no recompute)."""
    s = w.scaling
    A1, B1, A2, B2 = (p.detach().to(w.x1[0].dtype).requires_grad_(True) for p in (w.A1[0], w.B1[0], w.A2[0], w.B2[0]))
    x1, h, y2, g2 = w.x1[0], w.h[0], w.y2[0], w.g2[0]

    def run():
        for p in (A1, B1, A2, B2):
            p.grad = None
        xr = x1.detach().requires_grad_(True)
        h_out = h + ((xr @ A1) @ B1) * s              # original_layer(x) + lora(x), base output given
        y_out = y2 + ((h_out @ A2) @ B2) * s
        y_out.backward(g2)

    avg, _, _ = time_events(run, iters, warm=2, reps=3)
    return avg


def fused_site_measurement(w, iters=5):
    """SURVEY 8(f)-1 at the benchmark's fc1 site (M rows, 1024 -> 4736): `h = x W^T + b + s (x A) B`, `a = GELU(h)` as
    hipBLASLt GEMM + sam3_lora_fwd_act (k_t1 + k_t2<GELU>) against sam3_lora_linear_fwd (k_t1 + k_wext + k_fused_linear), interleaved
    in this process; the fused GEMM kernel alone from the library's in-situ timer, against the dense bf16 MFMA peak."""
    import ctypes
    from sam3_lora_amd import _ffi
    from sam3_lora_amd.functional import lora_fwd_, lora_linear_fwd_, pack_operands
    dev, M, s = w.dev, w.M, w.scaling
    if w.x1[0].dtype != torch.bfloat16 or w.rank > 32:
        return {"note": "bf16 activations and rank <= 32 only"}
    g = torch.Generator(device=dev).manual_seed(7)
    W = (torch.randn(D_HID, D_MODEL, device=dev, generator=g) / 32).bfloat16()
    b = (torch.randn(D_HID, device=dev, generator=g) * 0.1).bfloat16()
    x, A, B = w.x1[0], w.A1[0].detach(), w.B1[0].detach()
    blob = pack_operands(A, B, 0)
    h, a = torch.empty(M, D_HID, device=dev, dtype=torch.bfloat16), torch.empty(M, D_HID, device=dev, dtype=torch.bfloat16)

    def two_pass():
        torch.addmm(b, x, W.t(), out=h)
        lora_fwd_(x, A, B, h, s, 0, packed=blob, gelu_out=a)

    def fused():
        lora_linear_fwd_(x, W, b, A, B, s, 0, packed=blob, gelu=True, y_out=h, gelu_out=a)
    # Third contender: the strongest LIBRARY form of the same arithmetic -- hipBLASLt with K extended by the rank-r slots,
    # [x | t_hi t_lo t_hi 0] . [W | (sB)_hi (sB)_hi (sB)_lo 0]^T + b: the hi + lo products of the stand-alone kernels inside the frozen GEMM's fp32
    # accumulator, ONE rounding of h as in k_fused_linear.  hipBLASLt's GELU epilogue is the tanh approximation (the reference's nn.GELU is
    # the erf form, vitdet.py / timm Mlp), so GELU stays a pass of its own (torch's erf kernel here).  In production the LayerNorm would
    # write x at pitch 1088 and k_t1 the t slots; here x and the t slots are placed beforehand and k_t1's measured time is added.
    KX = D_MODEL + 64
    xext = torch.zeros(M, KX, device=dev, dtype=torch.bfloat16)
    Wfull = torch.zeros(D_HID, KX, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        xext[:, :D_MODEL] = x
        Wfull[:, :D_MODEL] = W
        t32 = x.float() @ A.float()                         # (placement only: the timed legs below never recompute it)
        t_hi = t32.bfloat16()
        t_lo = (t32 - t_hi.float()).bfloat16()
        r = A.shape[1]
        sB = (s * B.float()).t().contiguous()               # [N, r]
        b_hi = sB.bfloat16()
        b_lo = (sB - b_hi.float()).bfloat16()
        xext[:, D_MODEL:D_MODEL + r], Wfull[:, D_MODEL:D_MODEL + r] = t_hi, b_hi
        if D_MODEL + 3 * r <= KX:
            xext[:, D_MODEL + r:D_MODEL + 2 * r], Wfull[:, D_MODEL + r:D_MODEL + 2 * r] = t_lo, b_hi
            xext[:, D_MODEL + 2 * r:D_MODEL + 3 * r], Wfull[:, D_MODEL + 2 * r:D_MODEL + 3 * r] = t_hi, b_lo
        del t32
    h3, a3 = torch.empty_like(h), torch.empty_like(a)

    def kext_gemm():
        torch.addmm(b, xext, Wfull.t(), out=h3)

    def kext():
        torch.addmm(b, xext, Wfull.t(), out=h3)
        torch.ops.aten.gelu.out(h3, approximate="none", out=a3)

    def gemm_plain():
        torch.addmm(b, x, W.t(), out=h)
    t2, tf, tk, tkg, tg = [], [], [], [], []
    for _ in range(3):
        t2.append(time_events(two_pass, iters, warm=1, reps=1)[0])
        tf.append(time_events(fused, iters, warm=1, reps=1)[0])
        tk.append(time_events(kext, iters, warm=1, reps=1)[0])
        tkg.append(time_events(kext_gemm, iters, warm=1, reps=1)[0])
        tg.append(time_events(gemm_plain, iters, warm=1, reps=1)[0])
    fused()
    kext()
    torch.cuda.synchronize()
    # the two one-rounding forms agree to one bf16 rounding of each other (different fp32 summation orders inside the accumulators)
    dh = (h3.float() - h.float()).abs()
    kext_agree = {"h_max_abs_over_max": round(float(dh.max() / h.float().abs().max()), 6),
                  "h_elements_beyond_one_rounding": int((dh > 2.0 ** -7 * h.float().abs() + 1e-6).sum()),
                  "a_max_abs_over_max": round(float((a3.float() - a.float()).abs().max() / a.float().abs().max()), 6)}
    lib = _ffi.load()
    cap = 16
    lib.sam3_lora_prof_start(_ffi.STAGE_FUSED, cap)
    for _ in range(5):
        fused()
    us, st, dm = (ctypes.c_float * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
    n = lib.sam3_lora_prof_stop(us, st, dm, cap)
    k_us = sorted(us[i] for i in range(n))[n // 2] if n > 0 else None
    flop = 2.0 * M * D_MODEL * D_HID
    t1_us = None
    lib.sam3_lora_prof_start(_ffi.STAGE_T1, cap)
    for _ in range(5):
        fused()
    n1 = lib.sam3_lora_prof_stop(us, st, dm, cap)
    if n1 > 0:
        t1_us = sorted(us[i] for i in range(n1))[n1 // 2]
    med = lambda v: sorted(v)[1]
    out = {"M": M, "in": D_MODEL, "out": D_HID, "two_pass_us": round(sorted(t2)[1], 1), "fused_us": round(sorted(tf)[1], 1),
           "speedup": round(sorted(t2)[1] / sorted(tf)[1], 3),
           "library_k_extended": {
               "gemm_k1088_us": round(med(tkg), 1), "gemm_k1024_us": round(med(tg), 1), "gemm_plus_erf_gelu_pass_us": round(med(tk), 1),
               "k_t1_us_added": None if t1_us is None else round(t1_us, 1),
               "site_us": None if t1_us is None else round(med(tk) + t1_us, 1), "agreement_with_k_fused_linear": kext_agree,
               "what": "hipBLASLt (torch.addmm) over K = 1024 + 64: [x | t_hi t_lo t_hi 0] . [W | (sB)_hi (sB)_hi (sB)_lo 0]^T + b -- the same hi + lo "
                       "products and the same single rounding of h as k_fused_linear -- followed by an erf-GELU pass of its own (hipBLASLt's GELU "
                       "epilogue is the tanh form: not the reference's function), plus k_t1's in-situ time for the t slots (x and t are placed at "
                       "pitch 1088 beforehand: in production the LayerNorm and k_t1 would write them there)"},
           "what": "the fc1 -> GELU site: hipBLASLt GEMM + sam3_lora_fwd_act against sam3_lora_linear_fwd (the adapter inside the "
                   "frozen GEMM), interleaved rounds, HIP events on the current stream"}
    if k_us:
        out["roofline"] = {"bound": "mfma", "kernel": f"k_fused_linear [M={M},N={D_HID},K={D_MODEL}+64]", "avg_us": round(k_us, 1),
                           "achieved": round(flop / k_us / 1e6, 1), "peak": 2500.0, "unit": "TFLOP/s",
                           "frac": round(flop / k_us / 1e6 / 2500.0, 4),
                           "hbm_bytes_written": 2 * M * D_HID * 2,
                           "what": "frozen-GEMM FLOPs only (the rank-r K step and the GELU epilogue are overhead) over the in-situ kernel "
                                   "time, against the 2.5 PFLOP/s dense bf16 peak; PMC WRITE_SIZE = exactly 2 M N e "
                                   "(profiles/r04d_fc1_site_pmc_write.txt)"}
    return out


def trunk_step_bench(dev, batch, rank, steps, world, checkpoint=True):
    """The adapters in their real host: the SAM3 ViT-Det trunk (sam3_lora_amd/vit.py, 32 blocks, 1008^2 input,
    random init, frozen weights bf16) with root-API LoRA on fc1/fc2, one training step = forward with per-block
    activation checkpointing + backward + flat-buffer gradient exchange + AdamW on A/B.  Frozen GEMMs, SDPA,
    LayerNorm run on PyTorch-ROCm; the adapter arithmetic on the HIP path.  (Neck, text tower, DETR, losses are
    not part of this measurement -- they are not built yet.)"""
    import contextlib
    import io
    import lora_layers as L
    from sam3_lora_amd import vit as V
    from sam3_lora_amd.ddp import LoRAGradReducer
    torch.manual_seed(0)
    with torch.device(dev):
        model = V.sam3_vit(use_act_checkpoint=checkpoint)
    with contextlib.redirect_stdout(io.StringIO()):
        L.apply_lora_to_model(model, L.LoRAConfig(rank=rank, alpha=2 * rank, dropout=0.0, target_modules=["fc1", "fc2"],
                                                  apply_to_text_encoder=False, apply_to_detr_encoder=False,
                                                  apply_to_detr_decoder=False))
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, L.LoRALayer):
                m.lora_B.normal_(0, 0.02)
    V.to_training_layout(model)
    model.train()
    params = L.get_lora_parameters(model)
    red = LoRAGradReducer(params, bucket_bytes=8 << 20)
    opt = torch.optim.AdamW(params, lr=5e-5, weight_decay=0.01)
    g = torch.Generator(device=dev).manual_seed(1234)
    img = (torch.rand(batch, 3, 1008, 1008, device=dev, generator=g) * 2 - 1).bfloat16()
    tgt = torch.randn(batch, 1024, 72, 72, device=dev, generator=g).bfloat16()

    def step():
        red.zero_grad()
        feat = model(img)[0]
        loss = (feat.float() * tgt.float()).mean()
        loss.backward()
        red.finish()
        opt.step()
        return loss

    tw = time.perf_counter()
    step()
    torch.cuda.synchronize()
    warm_s = time.perf_counter() - tw
    slow = torch.tensor([1.0 if warm_s > 20.0 else 0.0], device=dev)
    if world > 1:
        dist.all_reduce(slow, op=dist.ReduceOp.MAX)
        dist.barrier()
    if slow.item() > 0:      # something is badly wrong on this box (e.g. ranks sharing a GPU): do not stall the bench
        steps = 1
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    n_lora = sum(p.numel() for p in params)
    out = dict(images_per_s=round(world * batch * steps / dt, 2), ms_per_step=round(dt / steps * 1e3, 2), steps=steps,
               loss_finite=bool(torch.isfinite(loss).item()), trainable_parameters=n_lora,
               peak_mem_gb=round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
               workload=f"SAM3 ViT-Det trunk only (32 blocks, 1024-d, 72x72 tokens), batch {batch}/GPU @ 1008^2, bf16 frozen "
                        f"weights, LoRA r={rank} on 64 MLP Linears, "
                        f"{'per-block activation checkpointing' if checkpoint else 'NO activation checkpointing'}, synthetic feature loss, "
                        f"AdamW on A/B, flat-buffer all-reduce")
    del model, opt, red, img, tgt
    torch.cuda.empty_cache()
    return out


def adapter_cpu_port(rank, seconds_budget=10.0):
    """numpy oracle of the adapter arithmetic alone (kind 'port') on a bounded sample: 1 image through the 64-Linear
    fwd + recompute + bwd schedule, fp32 -- the CPU counterpart of `adapter_path`."""
    import numpy as np
    from oracle import lora_oracle as O
    M, r, s = TOKENS, rank, 2.0
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(host_cores()[1])        # BLAS threads = the cores this process may use
    except Exception:
        pass
    rng = np.random.default_rng(0)
    x1 = rng.standard_normal((M, D_MODEL), dtype=np.float32)
    h = rng.standard_normal((M, D_HID), dtype=np.float32)
    y2 = rng.standard_normal((M, D_MODEL), dtype=np.float32)
    g2 = rng.standard_normal((M, D_MODEL), dtype=np.float32)
    gh = rng.standard_normal((M, D_HID), dtype=np.float32)
    g1 = rng.standard_normal((M, D_MODEL), dtype=np.float32)
    A1 = rng.uniform(-.25, .25, (D_MODEL, r)).astype(np.float32)
    B1 = (rng.standard_normal((r, D_HID)) * 2e-3).astype(np.float32)
    A2 = rng.uniform(-.25, .25, (D_HID, r)).astype(np.float32)
    B2 = (rng.standard_normal((r, D_MODEL)) * 2e-3).astype(np.float32)

    def block():
        nonlocal h, y2, gh, g1
        for _ in range(2):  # forward + recompute
            h += O.adapter_delta(x1, A1, B1, s, 0)
            y2 += O.adapter_delta(h, A2, B2, s, 0)
        gx, gA, gB = O.adapter_backward(g2, h, A2, B2, s, 0)
        gh += gx
        gx, gA, gB = O.adapter_backward(gh, x1, A1, B1, s, 0)
        g1 += gx

    block()  # warm
    t0 = time.perf_counter()
    n = 0
    while n < N_BLOCKS and time.perf_counter() - t0 < seconds_budget:
        block()
        n += 1
    dt = time.perf_counter() - t0
    per_img_s = dt / n * N_BLOCKS
    return dict(value=round(1.0 / per_img_s, 4), unit="images/s", kind="port", cores=host_cores()[1],
                sample=f"numpy fp32 oracle of the adapter arithmetic, 1 image (M={M}), {n}/{N_BLOCKS} ViT blocks timed, "
                       f"extrapolated to 32 blocks; {dt:.1f}s of CPU work")


def host_cores():
    """(threads torch uses now, cores this process may actually use): physical cores, capped by the CPU affinity mask
    and by the cgroup CPU quota (a container that sees 256 hardware threads may be allowed 16 CPUs of them; running 128
    OpenMP threads against that quota is several times slower than 16)."""
    usable = None
    try:
        import psutil
        usable = psutil.cpu_count(logical=False)
    except Exception:
        pass
    usable = int(usable or os.cpu_count() or 1)
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = min(usable, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return torch.get_num_threads(), usable


def model_setup(kind):
    """(config, resolution, source size) of the whole-step workload."""
    from sam3_lora_amd.sam3_image import SAM3_CONFIG, TINY_CONFIG
    if kind == "tiny":
        cfg = dict(TINY_CONFIG, text=dict(TINY_CONFIG["text"], vocab_size=49408, context_length=32))
        return cfg, 112, 128
    return dict(SAM3_CONFIG), 1008, 1024


def cpu_baseline(kind="sam3", seconds_budget=120.0, timed_steps=0):
    """SURVEY section 8(d) "CPU baseline beside it": the reference's own training step -- configs[0] = c1: minimal LoRA
    (rank 4 on the ViT MLPs, vision encoder only), B = 2 images, fp32, CPU, `torch.set_num_threads(physical cores)` --
    re-enacted by the oracle-side restatement: this library's PyTorch host model (sam3_image) with the adapters written as
    the reference writes them (oracle/lora_torch_cpu.py: base + ((x @ A) @ B) * s, torch autograd), matcher (twice per
    step, as the reference), loss wrapper, AdamW, per-block activation recompute.

    Protocol: c1 asks for 1 warm-up + >= 2 timed steps.  `timed_steps` > 0 (bench.py --cpu-steps N) runs exactly that.
    By default the sample is BOUNDED so that the bench line still arrives within minutes: a probe (the whole step on a
    trunk cut to 1 windowed + 1 global block, plus one more block of each kind timed alone) predicts the cost of a B = 2
    step; when warm-up + 2 timed steps fit `seconds_budget` they are run, otherwise ONE B = 2 step is timed (no warm-up:
    a CPU step of ~1.5 minutes is not first-touch dominated), otherwise the figure is the probe's extrapolation --
    `sample` says which."""
    from oracle.lora_torch_cpu import apply_reference_form_lora
    from sam3_lora_amd.sam3_data import SyntheticSegmentDataset, collate_fn_api
    from sam3_lora_amd.sam3_image import build_sam3_image_model
    from sam3_lora_amd.trainer import build_criterion, match_all_steps
    threads, phys = host_cores()
    torch.set_num_threads(phys)
    SAM3_CONFIG, res, src = model_setup(kind)
    B = 2
    ds = SyntheticSegmentDataset(2 * B, resolution=res, source=src)
    batches = [collate_fn_api([ds[k * B + i] for i in range(B)], dict_key="input", with_seg_masks=True)["input"] for k in range(2)]

    torch.manual_seed(0)
    model = build_sam3_image_model(device="cpu", eval_mode=False, config=SAM3_CONFIG, match_in_forward=True)
    n_adapted = apply_reference_form_lora(model, rank=4, alpha=8, targets=("fc1", "fc2"), only_under="vision_backbone")
    model.train()
    matcher, wrapper = build_criterion("local")
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01)
    trunk = model.backbone.vision_backbone.trunk
    all_blocks, all_global = list(trunk.blocks), list(trunk.full_attn_ids)
    windowed = [i for i in range(len(all_blocks)) if i not in all_global]

    def one_step(block_ids, k=0):
        """The whole step with the trunk reduced to ``block_ids`` (None = all blocks)."""
        ids = list(range(len(all_blocks))) if block_ids is None else list(block_ids)
        trunk.blocks = torch.nn.ModuleList([all_blocks[i] for i in ids])
        trunk.full_attn_ids = [len(ids) - 1] if block_ids is not None else all_global
        batch = batches[k & 1]
        t0 = time.perf_counter()
        out = model(batch)
        targets = [model.back_convert(t) for t in batch.find_targets]
        match_all_steps(matcher, out.output, targets)
        loss = wrapper(out, targets)["core_loss"]
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        trunk.blocks, trunk.full_attn_ids = torch.nn.ModuleList(all_blocks), all_global
        return dt, loss.item()

    what = (f"{'full 840M-parameter' if kind == 'sam3' else 'TINY-width (contract test)'} model, B = {B} synthetic images @ "
            f"{res}^2 (configs[0]), {n_adapted} rank-4 adapters in reference form, fwd + 2x matching + loss + bwd (per-block "
            f"recompute) + AdamW, fp32, {phys} threads")
    if timed_steps > 0:
        t_warm, _ = one_step(None, 0)
        ts = [one_step(None, 1 + i) for i in range(timed_steps)]
        t_full = sum(t for t, _ in ts) / len(ts)
        sample = (f"c1 protocol: 1 warm-up ({t_warm:.1f}s) + {timed_steps} timed steps ({', '.join(f'{t:.1f}s' for t, _ in ts)}); "
                  f"{what}; loss {ts[-1][1]:.2f}")
        torch.set_num_threads(threads)
        return dict(value=round(B / t_full, 5), unit="images/s", cores=phys, kind="port", s_per_step=round(t_full, 2),
                    batch=B, sample=sample)

    # bounded sample: the whole model with the trunk cut to (1 windowed + 1 global) block, plus the cost of one more
    # windowed / global block measured on the block alone (forward, recompute, backward -- the reference's checkpointing)
    from torch.utils.checkpoint import checkpoint as _ckpt
    w0, g0 = windowed[0], all_global[0]
    t_11, _ = one_step([w0, g0])
    side = res // SAM3_CONFIG["vit"]["patch_size"]
    probe_x = torch.randn(B, side, side, SAM3_CONFIG["vit"]["embed_dim"])

    def block_alone(i):
        x = probe_x.clone().requires_grad_(True)
        t0 = time.perf_counter()
        _ckpt(all_blocks[i], x, use_reentrant=False).sum().backward()
        return time.perf_counter() - t0
    Tw, Tg = block_alone(w0), block_alone(g0)
    predicted = t_11 + (len(windowed) - 1) * Tw + (len(all_global) - 1) * Tg
    if 3 * predicted <= seconds_budget:
        t_warm, _ = one_step(None, 0)
        ts = [one_step(None, 1 + i) for i in range(2)]
        t_full = sum(t for t, _ in ts) / 2
        sample = (f"c1 protocol: 1 warm-up ({t_warm:.1f}s) + 2 timed steps ({ts[0][0]:.1f}s, {ts[1][0]:.1f}s); {what}; "
                  f"loss {ts[-1][1]:.2f}")
    elif predicted <= seconds_budget:
        t_full, loss = one_step(None)
        sample = (f"ONE whole CPU training step timed, no warm-up (warm-up + 2 timed steps were predicted at {3 * predicted:.0f}s, over "
                  f"the {seconds_budget:.0f}s budget of a default bench run; `bench.py --cpu-steps 2` runs the c1 protocol in full): "
                  f"{what}; loss {loss:.2f}; {t_full:.1f}s of CPU work (predicted {predicted:.1f}s)")
    else:
        t_full = predicted
        sample = (f"extrapolated: one CPU training step of the whole model with the trunk cut to 1 windowed + 1 global block "
                  f"took {t_11:.1f}s; one more windowed / global block (forward + recompute + backward, timed alone) costs "
                  f"{Tw:.2f}s / {Tg:.2f}s -> {predicted:.1f}s for the {len(all_blocks)}-block step (over the {seconds_budget:.0f}s "
                  f"budget, not run whole); {what}")
    torch.set_num_threads(threads)
    return dict(value=round(B / t_full, 5), unit="images/s", cores=phys, kind="port", s_per_step=round(t_full, 2), batch=B,
                sample=sample)


# ============================================================================================ the full step ==
class FullStep:
    """One data-parallel rank of the training step of train_sam3_lora_native.py:887-943 on this library's SAM3 image
    model: synthetic batch (resident in HBM) -> forward (ViT trunk with the HIP adapter path, neck, text tower, fusion
    encoder, decoder, mask head) -> back_convert -> Hungarian matching of the final and every auxiliary output ->
    Sam3LossWrapper -> backward -> flat-buffer all-reduce of the A/B gradients -> AdamW on A/B."""

    def __init__(self, dev, batch, lora_rank, world, rank, dropout=0.0, act_checkpoint="auto", match_once=True,
                 bf16=True, kind="sam3"):
        import contextlib
        import io
        import lora_layers as L
        from sam3_lora_amd import vit as V
        from sam3_lora_amd.ddp import LoRAGradReducer
        from sam3_lora_amd.sam3_data import SyntheticSegmentDataset, collate_fn_api
        from sam3_lora_amd.sam3_image import build_sam3_image_model
        from sam3_lora_amd.trainer import build_criterion, move_to_device
        self.dev, self.batch, self.world, self.bf16 = dev, batch, world, bf16
        cfg, res, src = model_setup(kind)
        self.kind, self.res = kind, res
        self.model = build_sam3_image_model(device=str(dev), eval_mode=False, match_in_forward=not match_once, seed=0,
                                            config=cfg)
        with contextlib.redirect_stdout(io.StringIO()):
            L.apply_lora_to_model(self.model, L.LoRAConfig(
                rank=lora_rank, alpha=2 * lora_rank, dropout=dropout,
                target_modules=["q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2"], apply_to_vision_encoder=True,
                apply_to_text_encoder=True, apply_to_geometry_encoder=True, apply_to_detr_encoder=True,
                apply_to_detr_decoder=True, apply_to_mask_decoder=True))
        self.n_adapted = sum(isinstance(m, L.LoRALinear) for m in self.model.modules())
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():       # a warmed state (B != 0) so that all four gradient products are exercised
            for m in self.model.modules():
                if isinstance(m, L.LoRALayer):
                    m.lora_B.copy_(torch.randn(m.lora_B.shape, generator=g) * 0.02)
        self.model.to(dev)
        if bf16:
            V.to_training_layout(self.model)
        self.model.train()
        mode = {"on": True, "off": False}.get(act_checkpoint, "auto")
        if mode == "auto":          # one forward so that the policy sees the real token count; then decide
            self.ckpt = V.set_activation_checkpointing(self.model, "auto", batch=batch)
        else:
            self.ckpt = V.set_activation_checkpointing(self.model, mode, batch=batch)
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        self.reducer = LoRAGradReducer(self.params, bucket_bytes=8 << 20)
        from sam3_lora_amd.functional import direct_grad_accumulation
        self._direct = direct_grad_accumulation      # the kernels add straight into the reducer's flat buffer (scoped to backward)
        from sam3_lora_amd.trainer import make_adamw
        self.opt = make_adamw(self.params, lr=5e-5, weight_decay=0.01)         # torch's fused AdamW on the GPU (one kernel for the 128 tensors)
        self.matcher, self.wrapper = build_criterion("global" if world > 1 else "local")
        if match_once:
            self.model.set_prefetch_matcher(self.wrapper)
        ds = SyntheticSegmentDataset(2 * batch * world, resolution=res, source=src)
        self.batches = []
        for k in range(2):          # two resident batches, alternated
            idx = [k * batch * world + rank * batch + i for i in range(batch)]
            b = collate_fn_api([ds[i] for i in idx], dict_key="input", with_seg_masks=True)["input"]
            b = move_to_device(b, dev)
            if bf16:
                b.img_batch = b.img_batch.bfloat16()
            self.batches.append(b)
        self.n = 0
        self.last_loss = None
        from sam3_lora_amd.functional import repack_adapters
        self._repack = repack_adapters

    def step(self, timers=None, batch=None):
        from sam3_lora_amd.trainer import match_all_steps
        b = self.batches[self.n & 1] if batch is None else batch
        self.n += 1

        def mark(name):
            if timers is not None:
                torch.cuda.synchronize()
                timers.append((name, time.perf_counter()))
        mark("start")
        out = self.model(b)
        mark("forward")
        targets = [self.model.back_convert(t) for t in b.find_targets]
        match_all_steps(self.wrapper, out.output, targets)
        mark("matching (host LSAP)")
        loss = self.wrapper(out, targets)["core_loss"]
        mark("loss")
        self.reducer.zero_grad()
        with self._direct(True):
            loss.backward()
        if getattr(self, "backward_end_event", None) is not None:
            self.backward_end_event.record(torch.cuda.current_stream(self.dev))
        mark("backward")
        self.reducer.finish()
        from sam3_lora_amd import fp8 as _fp8
        if _fp8.fp8_enabled():              # the fp8 mode's trainer skips a step with non-finite gradients (trainer.NonFiniteStepGuard)
            from sam3_lora_amd.trainer import NonFiniteStepGuard
            if getattr(self, "_guard", None) is None or self._guard.optimizer is not self.opt:
                self._guard = NonFiniteStepGuard(self.opt, self.dev)
            self._guard.step()
        else:
            self.opt.step()
        self._repack(self.model)            # operand images of all adapters, batched (sam3_lora_pack_many)
        mark("exchange + AdamW")
        self.last_loss = loss
        return loss


def data_step_measurement(full, args, world, rank, timed):
    """SURVEY section 8f-4, VERDICT r2 item 6: the same whole training step with the DATA STEP IN THE LOOP instead of two
    resident batches -- every step pulls a fresh batch of synthetic samples (generated, resized, normalised, masks rasterised:
    sam3_data.synthetic_datapoint) from the rank's ShardedLoader, (a) through the worker pool (samples built by threads ahead
    of the step, collated, copied from pinned memory on a side stream) and (b) the reference's way, num_workers=0, inline."""
    from sam3_lora_amd.sam3_data import ShardedLoader, SyntheticSegmentDataset, collate_fn_api
    _, res, src = model_setup(full.kind)
    steps = max(3, min(args.steps, 6))
    workers = max(2, min(16, host_cores()[1]))
    collate = lambda smp: collate_fn_api(smp, dict_key="input", with_seg_masks=True)
    out = {}
    for name, nw, n_steps in (("worker_pool", workers, steps), ("inline_num_workers_0", 0, 2)):
        ds = SyntheticSegmentDataset(full.batch * world * (n_steps + 2), resolution=res, source=src)
        loader = ShardedLoader(ds, full.batch, collate, shuffle=False, rank=rank, world=world, num_workers=nw, prefetch=3,
                               device=full.dev if nw else None)
        it = iter(loader)

        def one():
            from sam3_lora_amd.trainer import move_to_device
            b = next(it)["input"]
            b = move_to_device(b, full.dev)
            if full.bf16:
                b.img_batch = b.img_batch.bfloat16()
            full.step(batch=b)
        one()
        one()
        loader.stall_s = 0.0
        dt = timed(one, n_steps)
        out[name] = {"images_per_s": round(world * full.batch * n_steps / dt, 2), "ms_per_step": round(dt / n_steps * 1e3, 2),
                     "steps": n_steps, "loader_threads": nw,
                     "step_waited_for_data_ms": round(loader.stall_s / n_steps * 1e3, 2) if nw else None}
        del it, loader
    out["what"] = ("whole training step with a fresh synthetic batch per step from sam3_data.ShardedLoader: `worker_pool` = samples "
                   "built by a thread pool + pinned async H2D ahead of the step; `inline_num_workers_0` = the reference's "
                   "DataLoader(num_workers=0) behaviour (train_sam3_lora_native.py:831)")
    return out


def rccl_info(world, dev):
    info = {"backend": dist.get_backend() if world > 1 else None, "device": torch.cuda.get_device_name(dev)}
    try:
        info["rccl_version"] = ".".join(map(str, torch.cuda.nccl.version()))
    except Exception as e:
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    ids = torch.tensor([dev.index if dev.index is not None else 0], device=dev)
    if world > 1:
        gathered = [torch.zeros_like(ids) for _ in range(world)]
        dist.all_gather(gathered, ids)
        info["rank_device_ids"] = [int(t.item()) for t in gathered]
    else:
        info["rank_device_ids"] = [int(ids.item())]
    return info


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node (one per GPU, RCCL) the way
    the driver would -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` -- and
    hand back its exit code.  A bare `--gpus 8` therefore can never run one rank and report `n_gpus: 1`."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch_under_torchrun(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a different job size",
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs an AMD GPU (the LoRA path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # test hook: BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo, so the N>1 code path (init, bucketed
    # side-stream all-reduce, barriers, MAX-over-ranks timing) can be exercised on a 1-GPU box.  Never set by the driver.
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    act_dtype = torch.bfloat16 if args.act_dtype == "bf16" else torch.float32
    out = None
    # ------------------------------------------------------------------ the headline: the whole training step
    full = None
    if not args.adapter_only:
        from sam3_lora_amd.fp8 import enable_fp8_frozen
        enable_fp8_frozen(bool(args.fp8_frozen))
        full = FullStep(dev, args.batch, args.rank, world, rank, dropout=args.dropout, act_checkpoint=args.act_checkpoint,
                        match_once=not args.match_twice, bf16=args.act_dtype == "bf16", kind=args.model)
        for _ in range(args.warmup):
            full.step()
        dt = timed(full.step, args.steps)
        finite = bool(torch.isfinite(full.last_loss).item()) and all(
            torch.isfinite(p.grad).all().item() for p in full.params[:8])
        info = rccl_info(world, dev)
        if rank == 0:
            ms = dt / args.steps * 1e3
            ips = world * args.batch / (dt / args.steps)
            out = {
                "metric": "training images/sec at 1024^2, SAM3-base r=%d (whole training step)" % args.rank,
                "value": round(ips, 2), "unit": "images/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.act_dtype, "data": "synthetic",
                "config": {"workload": workload_label(args, full.n_adapted, full.ckpt),
                           "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                           "grad_allreduce_bytes": full.reducer.nbytes, "finite": finite,
                           "loss": round(full.last_loss.item(), 4), "adapted_modules": full.n_adapted,
                           "trainable_parameters": sum(p.numel() for p in full.params)},
                # ~18 TFLOP per image and step -> ~140 images/s per GPU at the 2.5 PFLOP/s dense
                # bf16 MFMA peak; the whole step is MFMA/attention-bound, the adapter kernels HBM-bound (roofline below)
                "mfma_bound": {"images_per_s_per_gpu_at_peak": MFMA_BOUND_IPS, "frac": round(ips / world / MFMA_BOUND_IPS, 4),
                               "what": "whole-step images/s per GPU over the ~140 images/s MFMA bound of SURVEY 8(d)"},
                "distributed": info,
                "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
            }
        # phase breakdown of one extra (synchronising) step -- all ranks run it (it contains the exchange)
        marks = []
        full.step(timers=marks)
        if rank == 0:
            out["phases_ms"] = {name: round((t - marks[i][1]) * 1e3, 2) for i, (name, t) in enumerate(marks[1:])}
        if not args.no_overlap:
            ov = overlap_measurement(full, world)
            if rank == 0:
                out["exchange_overlap"] = ov
        if not args.no_data_step and not args.full_only and world == 1:     # auxiliary, N = 1 only: a rank failing alone here would
            try:                                                             # leave the others in a barrier
                dsm = data_step_measurement(full, args, world, rank, timed)
            except Exception as e:      # an auxiliary measurement must never cost the bench line
                dsm = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            if rank == 0:
                out["data_step"] = dsm
        if not args.fp8_frozen and args.act_dtype == "bf16" and not args.no_fused_ab:
            # SURVEY 8(f)-1 A/B inside the same process: the whole step with the fc1 -> GELU site through sam3_lora_linear_fwd
            # (the adapter inside the frozen GEMM) and through hipBLASLt + sam3_lora_fwd_act; whichever is NOT the default here
            from sam3_lora_amd.functional import fused_linear_enabled, set_fused_linear
            dflt = fused_linear_enabled()
            set_fused_linear(not dflt)
            try:
                for _ in range(2):
                    full.step()
                u_steps = max(3, min(args.steps, 6))
                dtu = timed(full.step, u_steps)
                if rank == 0:
                    out["fused_linear"] = {
                        "default_on": dflt, "this_variant_on": not dflt,
                        "value": round(world * args.batch * u_steps / dtu, 2), "unit": "images/s",
                        "ms_per_step": round(dtu / u_steps * 1e3, 3), "steps": u_steps, "loss": round(full.last_loss.item(), 4),
                        "what": "the same whole training step with the OTHER setting of SAM3_LORA_FUSED_LINEAR: on = fc1 + adapter + "
                                "bias + GELU as one MFMA kernel (sam3_lora_linear_fwd), off = hipBLASLt GEMM + sam3_lora_fwd_act"}
            finally:
                set_fused_linear(dflt)
        if not args.fp8_frozen and args.act_dtype == "bf16" and not args.no_fused_ab:
            # the two forms of SURVEY 8(f)-1 that are NOT the default, each switched on for a few steps of the same process: fc2's forward
            # through sam3_lora_linear_fwd, and the backward mirror gh = (ga W2 + s gt2 A2^T) GELU'(h) as one kernel (sam3_lora_linear_dgrad_act)
            from sam3_lora_amd.functional import set_knob
            vres = {}
            for knob, what in (("SAM3_LORA_FUSED_FC2", "fc2 forward (K = 4736, N = 1024) as one MFMA kernel instead of hipBLASLt + sam3_lora_fwd"),
                               ("SAM3_LORA_MIRROR", "fc2's input gradient x GELU'(h) as one MFMA kernel (sam3_lora_linear_dgrad_act) + the adapter's weight "
                                                    "gradients from the stored activation, instead of hipBLASLt + sam3_lora_bwd_act (which recomputes it)")):
                set_knob(knob, True)
                try:
                    for _ in range(2):
                        full.step()
                    v_steps = max(3, min(args.steps, 6))
                    dtv = timed(full.step, v_steps)
                    vres[knob] = {"ms_per_step": round(dtv / v_steps * 1e3, 3), "value": round(world * args.batch * v_steps / dtv, 2),
                                  "loss": round(full.last_loss.item(), 4), "what": what}
                except Exception as e:
                    vres[knob] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
                finally:
                    set_knob(knob, None)
            for _ in range(2):      # back on the default path before anything else is timed
                full.step()
            if rank == 0:
                out["fused_gemm_variants_not_default"] = vres
        if not args.no_fp8 and not args.fp8_frozen and args.act_dtype == "bf16":
            # BASELINE configs[4]'s mode on the same workload: frozen base GEMMs on the fp8 MFMA kernels
            loss_bf16 = full.last_loss.item()
            enable_fp8_frozen(True)
            try:
                for _ in range(2):
                    full.step()
                f_steps = max(3, min(args.steps, 6))
                dtf = timed(full.step, f_steps)
                # the fc1 -> GELU site of that mode through hipBLASLt's fp8 GEMM + sam3_lora_fwd_act_q8 instead of sam3_lora_linear_fwd_q8
                from sam3_lora_amd.functional import fused_linear_fp8_enabled, set_fused_linear_fp8
                dflt8 = fused_linear_fp8_enabled()
                set_fused_linear_fp8(not dflt8)
                try:
                    for _ in range(2):
                        full.step()
                    dtf2 = timed(full.step, f_steps)
                finally:
                    set_fused_linear_fp8(dflt8)
                if rank == 0:
                    out["fp8_frozen_w"] = {
                        "value": round(world * args.batch * f_steps / dtf, 2), "unit": "images/s",
                        "ms_per_step": round(dtf / f_steps * 1e3, 3), "steps": f_steps,
                        "fused_fc1_fp8": {"default_on": dflt8, "other_setting_ms_per_step": round(dtf2 / f_steps * 1e3, 3),
                                          "what": "the same fp8 step with the OTHER setting of SAM3_LORA_FUSED_LINEAR_FP8: on = fc1 as "
                                                  "sam3_lora_linear_fwd_q8 (e4m3 GEMM on the scaled fp8 MFMA + bf16 rank-r K step + GELU + fc2's "
                                                  "e4m3 image in one kernel), off = torch._scaled_mm + sam3_lora_fwd_act_q8"},
                        "loss": round(full.last_loss.item(), 4), "loss_bf16_build": round(loss_bf16, 4),
                        "finite": bool(torch.isfinite(full.last_loss).item()),
                        "what": "the same whole training step with the frozen Linears' GEMMs in fp8 (weights e4m3 per-tensor scale, "
                                "activations e4m3 / gradients e5m2 with delayed scaling -- written by the producing kernels (LayerNorm, the "
                                "GELU / GELU' adapter passes) where there is one, by the HIP quantiser otherwise; hipBLASLt fp8 MFMA "
                                "through torch._scaled_mm); LoRA branch bf16 / fp32 as before"}
            except (ValueError, FloatingPointError) as e:      # e.g. the Hungarian solver refusing a non-finite cost matrix
                if world > 1:
                    raise
                out["fp8_frozen_w"] = {"error": str(e)[:200], "finite": False, "loss_bf16_build": round(loss_bf16, 4)}
            enable_fp8_frozen(False)
        del full
        torch.cuda.empty_cache()
        if (not args.no_literal and args.rank == 16 and args.dropout == 0.0 and not args.fp8_frozen and args.act_dtype == "bf16"
                and args.model == "sam3"):
            # the reference's SHIPPED default (configs/full_lora_config.yaml:12-14: rank 32, alpha 64, dropout 0.1) on the same workload
            torch.cuda.reset_peak_memory_stats(dev)         # (its own peak, not the maximum over the sub-measurements before it)
            lit = FullStep(dev, args.batch, 32, world, rank, dropout=0.1, act_checkpoint=args.act_checkpoint,
                           match_once=not args.match_twice, bf16=True, kind=args.model)
            for _ in range(2):
                lit.step()
            l_steps = max(3, min(args.steps, 6))
            dtl = timed(lit.step, l_steps)
            if rank == 0:
                v = world * args.batch * l_steps / dtl
                out["literal_full_lora_config"] = {
                    "value": round(v, 2), "unit": "images/s", "ms_per_step": round(dtl / l_steps * 1e3, 3), "steps": l_steps,
                    "rank": 32, "alpha": 64, "dropout": 0.1, "vs_r16_line": round(v / out["value"], 4),
                    "finite": bool(torch.isfinite(lit.last_loss).item()), "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                    "what": "the same whole training step at configs/full_lora_config.yaml's literal adapter settings (r = 32, alpha = 64, "
                            "LoRA dropout 0.1 generated inside the kernels): hi + lo operand images as at r = 16 (64-wide), one pass over gy "
                            "(k_t3w with two rank tiles), fc2's input recomputed -- and masked -- inside the GELU' pass (not kept), fc1 + GELU "
                            "through sam3_lora_linear_fwd"}
            del lit
            torch.cuda.empty_cache()
        if (not args.no_fp32_layout and not args.fp8_frozen and args.act_dtype == "bf16" and args.model == "sam3" and world == 1):
            # the layout that MEETS north_star's 1e-3 on the logits (fp32 frozen tensors and activations, exact-fp32 adapter kernels:
            # tests/test_sam3_e2e.py::test_full_size_training_step_fp32_matches_reference, 1.3e-5 at full size), timed beside the bf16 one
            try:
                torch.cuda.reset_peak_memory_stats(dev)
                f32 = FullStep(dev, args.batch, args.rank, world, rank, dropout=args.dropout, act_checkpoint=args.act_checkpoint,
                               match_once=not args.match_twice, bf16=False, kind=args.model)
                f32.step()
                n32 = 2
                dt32 = timed(f32.step, n32)
                out["parity_layout_fp32"] = {
                    "value": round(world * args.batch * n32 / dt32, 2), "unit": "images/s", "ms_per_step": round(dt32 / n32 * 1e3, 2),
                    "steps": n32, "loss": round(f32.last_loss.item(), 4), "finite": bool(torch.isfinite(f32.last_loss).item()),
                    "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "vs_bf16_line": round(world * args.batch * n32 / dt32 / out["value"], 4),
                    "what": "the same whole training step in the exact-fp32 layout (the reference CLI's precision: fp32 frozen tensors and "
                            "activations on the fp32 MFMA / hipBLASLt fp32 GEMMs, exact-fp32 adapter kernels): the layout whose logits meet "
                            "north_star's 1e-3 against the reference at full size (measured 1.8e-5 over three AdamW steps, profiles/r06d_parity_full_fp32.json); "
                            "the bf16 line above sits at the reference's own autocast(bf16) deviation (2.7-3.4e-2 of max at full size, "
                            "profiles/r06d_parity_full_bf16.json)"}
                del f32
            except Exception as e:
                out["parity_layout_fp32"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not args.no_parity and args.model == "sam3":
            out["parity"] = parity_gates(dev, args)
        if args.full_only:
            if rank == 0:
                emit_line(out)
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            return
    # ------------------------------------------------------------------ the adapter path on its own (HBM roofline)
    w = Workload(dev, args.batch, args.rank, args.blocks, seed=1234 + rank, dropout=args.dropout, act_dtype=act_dtype)
    a_steps = args.steps if args.adapter_only else max(3, min(args.steps, 6))
    for _ in range(2):
        w.step()
    dta = timed(w.step, a_steps)
    if rank == 0:
        msa = dta / a_steps * 1e3
        ap = {"value": round(world * args.batch / (dta / a_steps), 2), "unit": "images/s", "ms_per_step": round(msa, 3),
              "steps": a_steps,
              "what": "the adapter path alone: %d ViT blocks x {fc1 1024->4736, fc2 4736->1024}, M=%d rows: one "
                      "sam3_lora_pack_many + 64 x sam3_lora_fwd + 64 x sam3_lora_fwd (checkpoint recompute) + 64 x sam3_lora_bwd "
                      "+ exchange; frozen GEMMs / attention / DETR / loss excluded" % (args.blocks, w.M)}
        if out is None:     # --adapter-only: the adapter path is the line
            out = {"metric": "images/sec through the LoRA adapter path (r=%d)" % args.rank, "value": ap["value"],
                   "unit": "images/s", "n_gpus": world, "steps": a_steps, "warmup": 2, "ms_per_step": ap["ms_per_step"],
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.act_dtype,
                   "data": "synthetic", "config": {"workload": ap["what"], "global_batch": world * args.batch,
                                                   "parallelism": "dp%d" % world}}
        out["adapter_path"] = ap
    for _ in range(2):
        w.step(recompute=False)
    nr_steps = 3
    tnr = timed(lambda: w.step(recompute=False), nr_steps)
    if rank == 0:
        out["adapter_path"]["no_recompute"] = {"value": round(world * args.batch * nr_steps / tnr, 2), "unit": "images/s",
                                               "ms_per_step": round(tnr / nr_steps * 1e3, 3)}
    rows = None
    if not args.no_roofline:
        rows = insitu_kernels(w)          # every rank (the instrumented steps contain the all-reduce)
    if rank == 0 and not args.no_roofline:
        ops = op_table(w, args.kernel_iters)
        dom = next(r_ for r_ in rows if r_["GBps"] is not None)      # largest share of the step's kernel time
        dimname = {"k_t1": "K", "k_t2": "N", "k_t3": "N"}.get(dom["kernel"], "dim")
        default_wl = args.batch == 8 and args.rank == 16 and args.act_dtype == "bf16" and args.blocks == N_BLOCKS
        tr = committed_traffic(dom["kernel"], (dom["dim"] + 127) // 128 if dom["kernel"] == "k_t2" else -1) if default_wl else None
        blk = ops[-1]       # north_star's quantity: the fused LoRA forward + backward of one block (fc1 + fc2), four C-ABI calls
        blk_tr = committed_block_traffic() if default_wl else None
        out["roofline"] = {"bound": "hbm", "kernel": f"fwd+bwd of one block (sam3_lora_fwd x 2 + sam3_lora_bwd x 2, M={w.M}, r={w.rank})",
                           "achieved": blk["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": blk["frac_of_peak"],
                           "traffic": blk_tr["hbm_bytes"] if blk_tr else None, "traffic_source": blk_tr["source"] if blk_tr else None,
                           "algorithmic_bytes": blk["algorithmic_bytes"], "avg_us": blk["avg_us"],
                           "what": "SURVEY 8(d) algorithmic bytes of the four calls (fc1 / fc2 forward and backward at the benchmark's M) over "
                                   "their measured time (HIP events around back-to-back calls on the stream they launch on); per call in `ops`",
                           "dominant_kernel": {"kernel": f"{dom['kernel']} [M={w.M},{dimname}={dom['dim']}]", "launches_timed": dom["launches"],
                                               "achieved": dom["GBps"], "unit": "GB/s", "frac": round(dom["GBps"] / HBM_PEAK_GBPS, 4),
                                               "traffic": tr["hbm_bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
                                               "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_us": dom["avg_us"],
                                               "what": "the stand-alone adapter kernel with the largest share of the adapter step's time, in situ"}}
        out["kernels"] = rows
        out["ops"] = ops
        e_, r_, M_ = w.x1[0].element_size(), w.rank, w.M
        fwd_b = lambda i, o: e_ * M_ * (i + 2 * o) + e_ * r_ * (i + o)
        bwd_b = lambda i, o: e_ * M_ * (o + i + 2 * i) + 4 * r_ * (i + o) * 2
        per_block = 2 * (fwd_b(D_MODEL, D_HID) + fwd_b(D_HID, D_MODEL)) + bwd_b(D_MODEL, D_HID) + bwd_b(D_HID, D_MODEL)
        step_bytes = per_block * args.blocks
        msa = out["adapter_path"]["ms_per_step"]
        out["roofline"]["step"] = {"algorithmic_bytes": step_bytes, "ms": msa,
                                   "achieved": round(step_bytes / (msa * 1e-3) / 1e9, 1), "unit": "GB/s",
                                   "frac": round(step_bytes / (msa * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                   "what": "all 256 C-ABI calls of the adapter path (fwd + recompute + bwd of 64 Linears)"}
        nr_ms = out["adapter_path"]["no_recompute"]["ms_per_step"]
        seq_bytes = (per_block - fwd_b(D_MODEL, D_HID) - fwd_b(D_HID, D_MODEL)) * args.blocks
        out["roofline"]["in_sequence"] = {"algorithmic_bytes": seq_bytes, "ms": nr_ms, "us_per_block": round(nr_ms * 1e3 / args.blocks, 2),
                                          "achieved": round(seq_bytes / (nr_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                          "frac": round(seq_bytes / (nr_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                          "what": "the same four calls per block in the ORDER a training step issues them (forward of all %d "
                                                  "blocks, then backward, each block on its own activations; operand packing and the "
                                                  "gradient exchange of the step included): the no-recompute adapter step over %d x the "
                                                  "block's algorithmic bytes.  `frac` above times each call repeated on its own, where every "
                                                  "in-place pass starts on the write-back of its predecessor (k_t2 at N = 4736: 144 us there, "
                                                  "124 us here; profiles/r06ad_op_kernel_split.json)" % (args.blocks, args.blocks)}
        try:
            tu = torch_unfused_block(w)
            ours = ops[-1]["avg_us"]
            out["torch_unfused_block"] = {"avg_us": round(tu, 1), "this_library_avg_us": ours, "speedup": round(tu / ours, 2),
                                          "what": "fc1+fc2 adapters fwd+bwd of one block as base + ((x@A)@B)*s under "
                                                  "torch.autograd, same activation dtype, same GPU"}
        except Exception as e:      # an auxiliary comparison must never cost the bench line
            out["torch_unfused_block"] = {"error": str(e)[:200]}
        try:
            if args.blocks == N_BLOCKS or args.model == "sam3":
                out["fused_linear_site"] = fused_site_measurement(w)
                if "roofline" in out["fused_linear_site"]:      # the real step's dominant OWN kernel (fc1's forward runs inside it there)
                    out["roofline"]["step_dominant"] = dict(out["fused_linear_site"]["roofline"], what=(
                        "the whole training step's dominant kernel of this library: fc1 + adapter + bias + GELU as one MFMA kernel (32 launches "
                        "per step); MFMA-bound -- frozen-GEMM FLOPs over the in-situ kernel time against the 2.5 PFLOP/s dense bf16 peak"))
        except Exception as e:
            out["fused_linear_site"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        out["roofline"]["operands"] = ("hi + lo bf16 pairs for A, B, t, gt (fp32 arithmetic on the bf16 activations; "
                                       "include/sam3_lora_amd.h)" if e_ == 2 and os.environ.get("SAM3_LORA_SINGLE_ROUND", "0") in ("", "0") and w.rank <= int(os.environ.get("SAM3_LORA_HL_MAX_RANK", "32") or 32) else "single-rounded bf16 / fp32")
    # the same kernels with single-rounded operands (round 2's arithmetic: SAM3_LORA_SINGLE_ROUND=1) -- what the hi + lo form costs
    if not args.no_roofline and args.rank <= 16 and act_dtype == torch.bfloat16 and os.environ.get("SAM3_LORA_SINGLE_ROUND", "0") in ("", "0"):
        from sam3_lora_amd import _ffi
        os.environ["SAM3_LORA_SINGLE_ROUND"] = "1"
        _ffi.load().sam3_lora_debug_reload_knobs()
        try:
            w.P1, w.P2 = [None] * w.blocks, [None] * w.blocks          # blobs of the other layout must not be reused
            for _ in range(2):
                w.step(recompute=False)
            tsr = timed(lambda: w.step(recompute=False), 3)
            rows_sr = insitu_kernels(w, steps=1)
            if rank == 0:
                k = next((r_ for r_ in rows_sr if r_["kernel"] == "k_t2" and r_["dim"] == D_HID), rows_sr[0])
                out["roofline"]["single_rounded_operands"] = {
                    "kernel": f"{k['kernel']} [M={w.M},N={k['dim']}]", "avg_us": k["avg_us"], "achieved": k["GBps"],
                    "frac": round(k["GBps"] / HBM_PEAK_GBPS, 4),
                    "adapter_path_no_recompute_ms": round(tsr / 3 * 1e3, 3),
                    "what": "SAM3_LORA_SINGLE_ROUND=1: A, B, t, gt rounded to bf16 once each (round 2's arithmetic) -- the price of "
                            "the hi + lo operands is the difference to the line above"}
        finally:
            os.environ.pop("SAM3_LORA_SINGLE_ROUND", None)
            _ffi.load().sam3_lora_debug_reload_knobs()
            w.P1, w.P2 = [None] * w.blocks, [None] * w.blocks
    if world > 1:
        dist.barrier()
    del w
    torch.cuda.empty_cache()
    if args.trunk:
        tr = trunk_step_bench(dev, args.batch, args.rank, args.trunk_steps, world)
        tr2 = trunk_step_bench(dev, args.batch, args.rank, args.trunk_steps, world, checkpoint=False)
        if rank == 0:
            out["trunk_step"] = tr
            out["trunk_step_no_checkpoint"] = tr2
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.model, timed_steps=args.cpu_steps)
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        out["adapter_path"]["cpu_port"] = adapter_cpu_port(args.rank)
    if rank == 0:
        emit_line(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def parity_gates(dev, args):
    """BASELINE.md section 3 "parity gates run with the timing", in this process on this GPU:
    (1) the adapter C-ABI calls (forward + backward, hi + lo bf16 kernels, the benchmark's rank) on 2048 rows of the fc1 / fc2 shapes
        against the numpy fp64 oracle (oracle/lora_oracle.py, pinned to the reference by tests/golden/adapter_*.npz): y / gx within one
        bf16 rounding, gA / gB within 3e-5 of max;
    (2) the training steps at the full model size in exactly the layout the line above times (bf16 frozen tensors, fp32 islands, fused
        fc1, hi + lo operands) against the REFERENCE's fp32 CPU steps of tests/golden/e2e_full.npz (tests/golden/make_e2e_golden.py full):
        every discrete decision of the first step identical, loss / curve / A-B gradients at fixed bars, logits / boxes / masks beside the
        reference's own autocast(bf16) deviation at that size (tests/golden/ref_autocast_bf16.json)."""
    import numpy as np
    gates = {}
    try:
        from oracle import lora_oracle as O
        from sam3_lora_amd.functional import lora_bwd_, lora_fwd_
        rng = np.random.default_rng(0)
        worst = {"y": 0.0, "gx": 0.0, "gA": 0.0, "gB": 0.0, "elements_beyond_one_rounding": 0}
        for fin, fout in ((D_MODEL, D_HID), (D_HID, D_MODEL)):
            M, r, s = 2048, args.rank, 2.0
            x = O.bf16_round(rng.standard_normal((M, fin)).astype(np.float32))
            gy = O.bf16_round(rng.standard_normal((M, fout)).astype(np.float32))
            base = O.bf16_round(rng.standard_normal((M, fout)).astype(np.float32))
            gxb = O.bf16_round(rng.standard_normal((M, fin)).astype(np.float32))
            A = (rng.uniform(-1, 1, (fin, r)) / r ** 0.5).astype(np.float32)
            B = (rng.standard_normal((r, fout)) * 0.02).astype(np.float32)
            want_y = base + O.adapter_delta(x, A, B, s, 0, acc_dtype=np.float64)
            gx_l, gA_w, gB_w = O.adapter_backward(gy, x, A, B, s, 0, acc_dtype=np.float64)
            t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
            dA, dB = t(A), t(B)
            y = t(base, torch.bfloat16)
            tT = lora_fwd_(t(x, torch.bfloat16), dA, dB, y, s, 0, save_t=True)
            gx = t(gxb, torch.bfloat16)
            gA, gB = torch.zeros_like(dA), torch.zeros_like(dB)
            lora_bwd_(t(gy, torch.bfloat16), t(x, torch.bfloat16), tT, dA, dB, gx, gA, gB, s, 0)
            for name, got, ref in (("y", y.float().cpu().numpy(), want_y), ("gx", gx.float().cpu().numpy(), gxb + gx_l)):
                err = np.abs(got.astype(np.float64) - ref)
                worst[name] = max(worst[name], float(err.max() / np.abs(ref).max()))
                worst["elements_beyond_one_rounding"] += int((err > 2.0 ** -8 * np.abs(ref) + 3e-5 * np.abs(ref).max()).sum())
            worst["gA"] = max(worst["gA"], float(np.abs(gA.cpu().numpy() - gA_w).max() / np.abs(gA_w).max()))
            worst["gB"] = max(worst["gB"], float(np.abs(gB.cpu().numpy() - gB_w).max() / np.abs(gB_w).max()))
        worst["pass"] = bool(worst["elements_beyond_one_rounding"] == 0 and worst["gA"] < 3e-5 and worst["gB"] < 3e-5)
        worst["what"] = ("sam3_lora_fwd / sam3_lora_bwd at r = %d on 2048 rows of the fc1 and fc2 shapes against the fp64 oracle: max-abs error over "
                         "max|ref| per tensor; bars: every y / gx element within one bf16 rounding, gA / gB 3e-5" % args.rank)
        gates["adapter_vs_oracle"] = worst
    except Exception as e:
        gates["adapter_vs_oracle"] = {"error": f"{type(e).__name__}: {str(e)[:300]}", "pass": False}
    try:
        sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
        import test_sam3_e2e as T
        rec = T._full_size_step("bf16")
        yard = T._yardstick("full")
        sm, checks = T._full_bf16_verdict(rec, yard)        # the bars of test_full_size_training_step_bf16_layout_against_reference
        sm["losses"] = [round(v, 6) for v in rec["losses"]]
        yard_short = {k: v for k, v in yard.items() if k != "samples"}
        gates["full_size_step_vs_reference"] = dict(sm, checks=checks, reference_autocast_bf16_vs_its_fp32=yard_short, **{"pass": bool(all(checks.values()))}, what=(
            "the reference's fp32 CPU training steps at the REAL model size (tests/golden/e2e_full.npz: depth 32, 1008^2, 64 adapters, one image, "
            "%d AdamW steps) re-run here in the layout this line times.  Held strictly: the fixture's ground-truth boxes leave each of the first "
            "step's 12 discrete decisions (Hungarian assignment of the final + 5 auxiliary outputs and of the 5 auxiliary one-to-many twins, the "
            "final twin's threshold matches) a margin (tests/golden/margins.py, e2e_boxes.json), so `decisions_differing` must be empty; first-step "
            "total and every step of the loss curve within %g; worst A/B gradient within twice this build's measured deviation; logits / boxes / "
            "presence (max-abs error over max|ref|) within the reference's own autocast(bf16) deviation at that size (largest of three image "
            "samples), masks within twice it" % (len(rec["losses"]), T.BF16_CURVE_BAR)))
        torch.cuda.empty_cache()
    except Exception as e:
        gates["full_size_step_vs_reference"] = {"error": f"{type(e).__name__}: {str(e)[:300]}", "pass": False}
    gates["pass"] = bool(all(v.get("pass") for v in gates.values() if isinstance(v, dict)))
    return gates


def exchange_alone_measurement(full):
    """N = 1: hardware evidence for the side-stream mechanics on the one GPU there is.  A one-rank RCCL communicator is created and
    the reducer is told to issue its collectives anyway (``LoRAGradReducer.run_alone``: a 1-rank all-reduce is the identity, but it is
    a real RCCL launch on the side HIP stream, ordered by the same events): where each bucket's all-reduce starts and ends relative to
    the END of backward on the GPU's clock, how many collectives a step issues, and the step time with and without them (interleaved).
    No bytes cross xGMI here -- what N > 1 adds is the transfer time inside each all-reduce, not the launch positions."""
    red = full.reducer
    made_group = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=full.dev)
            made_group = True

        def run(n, alone):
            red.run_alone = alone
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                full.step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        run(1, True)                                    # communicator warm-up (the first collective builds its channels)
        with_x, without = [], []
        for _ in range(3):                              # interleaved: box drift does not land on one side
            without.append(run(2, False))
            with_x.append(run(2, True))
        red.run_alone = True
        red.trace = True
        full.backward_end_event = torch.cuda.Event(enable_timing=True)
        full.step()
        torch.cuda.synchronize()
        red.trace = False
        ev = full.backward_end_event
        full.backward_end_event = None
        buckets = [{"bucket": b, "origin": o, "bytes": 4 * (red.buckets[b][1] - red.buckets[b][0]),
                    "start_ms_after_backward_end": round(ev.elapsed_time(e0), 3), "end_ms_after_backward_end": round(ev.elapsed_time(e1), 3)}
                   for (b, e0, e1, o) in red.launch_events]
        log = list(red.launch_log)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(red.flat)
        torch.cuda.synchronize()
        alone_ms = (time.perf_counter() - t0) / 10 * 1e3
        med = lambda v: sorted(v)[len(v) // 2]
        return {"world": 1, "n_gpus_note": "N = 1: one-rank RCCL communicator, collectives issued although there is no peer (launch positions "
                                           "and stream ordering are real, transfer time is not)",
                "step_ms_with_exchange": round(med(with_x), 2), "step_ms_without": round(med(without), 2),
                "step_ms_delta": round(med(with_x) - med(without), 3), "samples_with": [round(v, 2) for v in with_x],
                "samples_without": [round(v, 2) for v in without], "allreduce_alone_ms": round(alone_ms, 3), "bytes": red.nbytes,
                "collectives_per_step": len(buckets), "launch_log": log, "buckets_rank0": buckets,
                "buckets_started_before_backward_end": sum(1 for b in buckets if b["start_ms_after_backward_end"] < 0)}
    except Exception as e:      # an auxiliary measurement must never cost the bench line
        return {"world": 1, "error": f"{type(e).__name__}: {str(e)[:300]}"}
    finally:
        red.run_alone = False
        red.trace = False
        if made_group:
            try:
                dist.destroy_process_group()
            except Exception:
                pass


def overlap_measurement(full, world):
    """How much of the A/B-gradient exchange hides behind the backward: the step timed with the bucketed side-stream
    all-reduce launched from the gradient hooks (overlapped) against the same step with one blocking all-reduce after
    the backward (exposed), and the exchange on its own.  With one rank there is nothing to exchange: reported as such."""
    if world == 1:
        return exchange_alone_measurement(full)
    red = full.reducer

    def run(n):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            full.step()
        torch.cuda.synchronize()
        dist.barrier()
        return (time.perf_counter() - t0) / n * 1e3

    overlapped = run(3)
    # where each bucket's all-reduce sits relative to the END of backward on the GPU's clock (HIP events on the side stream and on
    # the compute stream): negative start = launched while backward was still running, i.e. hidden behind it
    buckets = None
    if red.overlap:
        red.trace = True
        full.backward_end_event = torch.cuda.Event(enable_timing=True)
        full.step()
        torch.cuda.synchronize()
        red.trace = False
        ev = full.backward_end_event
        full.backward_end_event = None
        buckets = [{"bucket": b, "origin": o, "bytes": 4 * (red.buckets[b][1] - red.buckets[b][0]),
                    "start_ms_after_backward_end": round(ev.elapsed_time(e0), 3), "end_ms_after_backward_end": round(ev.elapsed_time(e1), 3)}
                   for (b, e0, e1, o) in red.launch_events]
    saved = red.overlap
    red.overlap = False
    exposed = run(3)
    red.overlap = saved
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dist.all_reduce(red.flat)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / 10 * 1e3
    t = torch.tensor([overlapped, exposed, alone], device=red.flat.device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    o, e, a = (float(v) for v in t.tolist())
    return {"world": world, "step_ms_overlapped": round(o, 2), "step_ms_exposed": round(e, 2),
            "allreduce_alone_ms": round(a, 3), "bytes": red.nbytes,
            "hidden_ms": round(max(e - o, 0.0), 3), "buckets_rank0": buckets,
            "collectives_per_step": len(red.buckets)}


if __name__ == "__main__":
    main()
