"""What the host gives a CPU baseline: visible cores, affinity, cgroup quota, torch threads, one fp32 GEMM rate."""
import os, time, torch
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
try:
    import psutil
    print("psutil physical", psutil.cpu_count(logical=False), "logical", psutil.cpu_count())
except Exception as e:
    print("psutil", e)
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    a, b = torch.randn(4096, 4096), torch.randn(4096, 4096)
    (a @ b)
    t0 = time.perf_counter()
    for _ in range(3):
        (a @ b)
    dt = (time.perf_counter() - t0) / 3
    print(f"threads {nt}: 4096^3 fp32 GEMM {dt*1e3:.1f} ms = {2*4096**3/dt/1e12:.2f} TFLOP/s")
