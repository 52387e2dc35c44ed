import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for b in (1, 2, 4):
    for ck in (True, False):
        r = bench.trunk_step_bench(dev, b, 16, 5, 1, checkpoint=ck)
        print(f"batch {b} checkpoint={ck}: {r['images_per_s']} img/s {r['ms_per_step']} ms/step", flush=True)
