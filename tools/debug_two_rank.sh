#!/bin/bash
# 2-rank toy CLI on one GPU (gloo), direct_grad on/off: max |rank0 - rank1| over the adapter tensors
cd $GRAFT_REPO_ROOT
for DG in true false; do
T=/tmp/tr_$DG; rm -rf $T; mkdir -p $T
python - <<PY
import json, yaml, sys
sys.path.insert(0, "tests")
import test_trainer as TT
cfg = json.loads(json.dumps(TT.BASE_CFG)); cfg["output"]["output_dir"] = "$T/out"; cfg["toy_no_valid"] = True
cfg["engine"] = {"direct_grad": $( [ $DG = true ] && echo True || echo False )}
open("$T/cfg.yaml", "w").write(yaml.safe_dump(cfg))
PY
for r in 0 1; do
RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 SAM3_LORA_DIST_BACKEND=gloo PYTHONPATH=tests:. TOY_DUMP=$T/rank$r.pt \
  python tests/toy_cli.py --config $T/cfg.yaml --model-builder toy_sam3:model_builder --data-builder toy_sam3:data_builder > $T/log$r.txt 2>&1 &
done
wait
python - <<PY
import torch
a, b = (torch.load("$T/rank%d.pt" % r) for r in range(2))
d = {k: (a[k] - b[k]).abs().max().item() for k in a}
bad = {k: v for k, v in d.items() if v > 0}
print("direct_grad=$DG: tensors", len(d), "differing", len(bad), "max diff", max(d.values()), "max |a|", max(v.abs().max().item() for v in a.values()))
print(list(bad.items())[:4])
PY
tail -3 $T/log0.txt
done
