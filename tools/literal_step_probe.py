"""Debug probe: the literal full_lora_config step (r = 32, dropout 0.1): two bf16 steps, then the fp8 frozen-GEMM mode for four;
prints the loss and the non-finite adapter gradients / parameters per step.  SAM3_LORA_AMD_LIB selects the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from sam3_lora_amd.fp8 import enable_fp8_frozen
    dev = torch.device("cuda:0")
    full = bench.FullStep(dev, 8, int(os.environ.get("PROBE_RANK", "32")), 1, 0, dropout=float(os.environ.get("PROBE_DROP", "0.1")))
    for step in range(6):
        if step == 2:
            enable_fp8_frozen(True)
        try:
            full.step()
            torch.cuda.synchronize()
            loss = full.last_loss.item()
        except Exception as e:  # noqa: BLE001
            loss = "EXC %s" % (str(e)[:60],)
        gbad = [n for n, p in full.model.named_parameters()
                if (p.grad is not None and not torch.isfinite(p.grad).all()) or (p.requires_grad and not torch.isfinite(p).all())]
        print("step", step, "fp8" if step >= 2 else "bf16", "loss", loss, "non-finite", len(gbad), gbad[:3], flush=True)
        if isinstance(loss, str):
            break


main()
