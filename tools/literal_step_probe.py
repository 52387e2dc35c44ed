"""Debug probe: the literal full_lora_config step (r = 32, dropout 0.1): PROBE_BF16_STEPS (2) bf16 steps, then the fp8 frozen-GEMM
mode for PROBE_FP8_STEPS (4); prints the loss, the non-finite adapter gradients / parameters and the fp8 roles whose delayed-scaling
state (amax lines, scale) is non-finite or zero, per step.  SAM3_LORA_AMD_LIB selects the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from sam3_lora_amd.fp8 import enable_fp8_frozen
    dev = torch.device("cuda:0")
    full = bench.FullStep(dev, 8, int(os.environ.get("PROBE_RANK", "32")), 1, 0, dropout=float(os.environ.get("PROBE_DROP", "0.1")))
    from sam3_lora_amd import fp8 as F8
    nb, nf = int(os.environ.get("PROBE_BF16_STEPS", "2")), int(os.environ.get("PROBE_FP8_STEPS", "4"))
    names = {id(p): n for n, p in full.model.named_parameters()}

    def fp8_state():
        bad = []
        for key, (_, st) in list(F8._WEIGHTS.items()):
            for role, q in (("x", st.qx), ("g", st.qg)):
                if q.amax is None:
                    continue
                a, sc = q.amax.float(), q.scale.float()
                if not torch.isfinite(a).all() or not torch.isfinite(sc).all() or (sc == 0).any():
                    bad.append((names.get(key, "?"), role, a.max().item(), sc.item()))
        return bad

    for step in range(nb + nf):
        if step == nb:
            enable_fp8_frozen(True)
        try:
            full.step()
            torch.cuda.synchronize()
            loss = full.last_loss.item()
        except Exception as e:  # noqa: BLE001
            loss = "EXC %s" % (str(e)[:60],)
        gbad = [n for n, p in full.model.named_parameters()
                if (p.grad is not None and not torch.isfinite(p.grad).all()) or (p.requires_grad and not torch.isfinite(p).all())]
        sbad = fp8_state() if step >= nb else []
        print("step", step, "fp8" if step >= nb else "bf16", "loss", loss, "non-finite", len(gbad), gbad[:3], "fp8 roles", len(sbad), sbad[:4], flush=True)
        if isinstance(loss, str):
            break


main()
