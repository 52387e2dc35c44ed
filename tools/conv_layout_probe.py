#!/usr/bin/env python3
"""Memory-format probe for the convolutional tail of the model (neck 4x level, pixel decoder, mask heads) on MI355X:
what layout MIOpen's kernels hand back for NCHW / channels_last inputs, what the conversions cost, and what GroupNorm
does with a channels_last input.  Shapes of the 288^2 level at batch 8, bf16, frozen weights (input gradient only)."""
import torch
import torch.nn.functional as F

dev = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n


def strides(t):
    return "NHWC" if t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous() else \
           ("NCHW" if t.is_contiguous() else str(t.stride()))


B, C, H = 8, 256, 288
conv = torch.nn.Conv2d(C, C, 3, 1, 1).to(dev, torch.bfloat16).requires_grad_(False)
conv1 = torch.nn.Conv2d(C, C, 1).to(dev, torch.bfloat16).requires_grad_(False)
gn = torch.nn.GroupNorm(8, C).to(dev, torch.bfloat16).requires_grad_(False)
for fmt in (torch.contiguous_format, torch.channels_last):
    x = torch.randn(B, C, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=fmt).requires_grad_(True)
    y = conv(x)
    print(f"input {strides(x)}: conv3x3 out {strides(y)}; conv1x1 out {strides(conv1(x))}; groupnorm out {strides(gn(x))}; "
          f"relu out {strides(F.relu(y))}; nearest-upsample out {strides(F.interpolate(x[:, :, :144, :144], size=(H, H)))}")
    g = torch.randn_like(y)

    def fb(m):
        def run():
            o = m(x)
            o.backward(g if o.shape == g.shape else torch.ones_like(o))
            x.grad = None
        return run
    for name, m in (("conv3x3", conv), ("conv1x1", conv1), ("groupnorm", gn), ("gn+relu", lambda t: F.relu(gn(t)))):
        tf = timeit(lambda: m(x.detach()))
        tb = timeit(fb(m))
        print(f"   {name:10s} fwd {tf:7.3f} ms   fwd+bwd {tb:7.3f} ms")
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        o = F.relu(gn(conv(x)))
        o.backward(g)
        torch.cuda.synchronize()
    rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 50]
    rows.sort(key=lambda r: -r[2])
    for k, c, t in rows[:14]:
        print(f"      {t / 1e3:8.3f} ms  x{c:<3d} {k[:110]}")

# AdamW on 128 small tensors: foreach vs fused
ps = [torch.randn(n, 16, device=dev, requires_grad=True) for n in (1024, 4736) * 64]
for p in ps:
    p.grad = torch.randn_like(p)
for kw in (dict(), dict(fused=True)):
    opt = torch.optim.AdamW(ps, lr=1e-4, weight_decay=0.01, **kw)
    t = timeit(lambda: opt.step(), n=20)
    print(f"AdamW {kw or 'foreach (default)'}: {t:.3f} ms / step (device timeline)")
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        opt.step()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"      host enqueue {((t1 - t0) / 20) * 1e3:.3f} ms / step")
