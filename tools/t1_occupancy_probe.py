import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for b in (8, 12, 16, 20, 24, 32):
    w = bench.Workload(dev, b, 16, 8, seed=1)
    for pad in ("", "0"):
        os.environ.pop("SAM3_LORA_T1_LDS_PAD", None) if pad == "" else os.environ.__setitem__("SAM3_LORA_T1_LDS_PAD", pad)
        w.step(); rows = bench.insitu_kernels(w, steps=2)
        t1 = {r["dim"]: (r["avg_us"], r["GBps"]) for r in rows if r["kernel"] == "k_t1"}
        print(f"batch {b} pad {pad:6s} k_t1@4736 {t1.get(4736)}  @1024 {t1.get(1024)}", flush=True)
    del w; torch.cuda.empty_cache()
