import os, sys, time, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from sam3_lora_amd import functional as Fn
dev = "cuda"
M, fin, fout, r = 256, 1024, 4736, 16
x = torch.randn(M, fin, device=dev).bfloat16(); y = torch.randn(M, fout, device=dev).bfloat16()
gy = torch.randn(M, fout, device=dev).bfloat16(); gx = torch.zeros(M, fin, device=dev).bfloat16()
A = torch.randn(fin, r, device=dev); B = torch.randn(r, fout, device=dev); gA = torch.zeros_like(A); gB = torch.zeros_like(B)
pk = Fn.pack_operands(A, B, 0)
for name, fn in (("lora_fwd_ (save_t, packed)", lambda: Fn.lora_fwd_(x, A, B, y, 2.0, 0, save_t=True, packed=pk)),
                 ("lora_bwd_ (packed)", lambda: Fn.lora_bwd_(gy, x, None, A, B, gx, gA, gB, 2.0, 0, accumulate=True, packed=pk)),
                 ("torch F.linear tiny", lambda: torch.nn.functional.linear(x, y[:64, :1024]))):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:32s} host {1e6*(t1-t0)/n:6.1f} us/call   (incl. queue back-pressure if GPU-bound: total {1e6*(time.perf_counter()-t0)/n:6.1f})")
