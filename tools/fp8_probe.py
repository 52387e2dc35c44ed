"""Does torch._scaled_mm run fp8 (OCP e4m3 / e5m2) GEMMs on this GPU, how exact, how fast against bf16?
Shapes: the SAM3 trunk's frozen Linears at batch 8 (M = 41,472)."""
import time
import torch

dev = "cuda:0"
torch.manual_seed(0)
print(torch.cuda.get_device_name(0), torch.__version__)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def q(t, dt):
    amax = t.abs().max().float()
    scale = amax / torch.finfo(dt).max
    return (t.float() / scale).to(dt), scale.reshape(1) if False else scale


for M, K, N in ((41472, 1024, 4736), (41472, 4736, 1024), (41472, 1024, 3072), (41472, 1024, 1024)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.03
    ref = x.float() @ w.float().t()
    t_bf = bench(lambda: torch.nn.functional.linear(x, w))
    line = f"M={M} K={K} N={N}: bf16 {t_bf:8.1f} us ({2*M*K*N/t_bf/1e6:6.0f} TF/s)"
    for adt, wdt in ((torch.float8_e4m3fn, torch.float8_e4m3fn), (torch.float8_e5m2, torch.float8_e4m3fn)):
        try:
            xq, sx = q(x, adt)
            wq, sw = q(w, wdt)
            f = lambda: torch._scaled_mm(xq, wq.t(), scale_a=sx, scale_b=sw, out_dtype=torch.bfloat16)
            y = f()
            err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
            t = bench(f)
            line += f" | {str(adt)[6:]}x{str(wdt)[6:]} {t:8.1f} us ({2*M*K*N/t/1e6:6.0f} TF/s) err {err:.3e}"
        except Exception as e:
            line += f" | {str(adt)[6:]}x{str(wdt)[6:]} FAILED {type(e).__name__}: {str(e)[:120]}"
    # cost of the scaled cast in plain torch ops
    sxv = x.abs().max().float() / 448
    t_cast = bench(lambda: (x.float() / sxv).to(torch.float8_e4m3fn))
    line += f" | torch cast {t_cast:.1f} us"
    print(line)
