import os, sys, json, tempfile, yaml, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "tests")]
from sam3_lora_amd import trainer as T
import toy_sam3
from test_trainer import BASE_CFG
d = tempfile.mkdtemp()
cfg = json.loads(json.dumps(BASE_CFG)); cfg["output"]["output_dir"] = d + "/out"; cfg["training"]["num_epochs"] = 1; cfg["toy_train_batches"] = 40
cfg["lora"]["dropout"] = 0.1
open(d + "/c.yaml", "w").write(yaml.safe_dump(cfg))
tr = T.SAM3TrainerNative(d + "/c.yaml", model_builder=toy_sam3.model_builder, data_builder=toy_sam3.data_builder, bf16_frozen=True, act_checkpoint="off")
mem = []
for ep in range(6):
    r = tr.train()
    torch.cuda.synchronize()
    mem.append((round(r["history"][-1]["train_loss"], 4), torch.cuda.memory_allocated() >> 10, torch.cuda.memory_reserved() >> 20))
print(mem)
assert mem[-1][1] <= mem[1][1] * 1.05 + 64, "allocated memory grows"
print("soak ok")
