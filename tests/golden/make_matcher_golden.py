#!/usr/bin/env python3
"""Golden vectors for the Hungarian matcher: runs the REFERENCE
``sam3.train.matcher.BinaryHungarianMatcherV2`` (imported from /root/reference, CPU) on seeded synthetic
predictions/targets and stores inputs + returned index tensors in ``matcher_cases.npz``.  Build container only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)
import numpy as np
import torch
import sam3_manifest

from matcher_case_defs import CASES


def make_inputs(name):
    kw, B, Q, nb, repeats, rb, uo, ut = CASES[name]
    g = torch.Generator().manual_seed(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    Bq = B * rb
    logits = torch.randn(Bq, Q, 1, generator=g)
    cxcy = torch.rand(Bq, Q, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(Bq, Q, 2, generator=g) * 0.3 + 0.05
    boxes = torch.cat([cxcy, wh], -1)
    T = max(max(nb), 1)
    tb = torch.cat([torch.rand(B, T, 2, generator=g) * 0.6 + 0.2, torch.rand(B, T, 2, generator=g) * 0.3 + 0.05], -1)
    for b, n in enumerate(nb):
        tb[b, n:] = 0
    out_valid = (torch.rand(Bq, Q, generator=g) > 0.3) if uo else None
    tgt_valid = None
    if ut:
        tgt_valid = torch.zeros(B, T, dtype=torch.bool)
        for b, n in enumerate(nb):
            tgt_valid[b, :n] = torch.rand(n, generator=g) > 0.25
    return logits, boxes, tb, torch.tensor(nb, dtype=torch.long), out_valid, tgt_valid


def main():
    sam3_manifest._install_stubs()
    from sam3.train.matcher import BinaryHungarianMatcherV2
    out = {}
    for name, (kw, B, Q, nb, repeats, rb, uo, ut) in CASES.items():
        logits, boxes, tb, nbt, ov, tv = make_inputs(name)
        m = BinaryHungarianMatcherV2(**kw)
        bi, si, ti = m({"pred_logits": logits, "pred_boxes": boxes}, {"num_boxes": nbt, "boxes_padded": tb},
                       repeats=repeats, repeat_batch=rb, out_is_valid=ov, target_is_valid_padded=tv)
        out[f"{name}/logits"], out[f"{name}/boxes"], out[f"{name}/tgt"] = logits.numpy(), boxes.numpy(), tb.numpy()
        out[f"{name}/num_boxes"] = nbt.numpy()
        if ov is not None:
            out[f"{name}/out_valid"] = ov.numpy()
        if tv is not None:
            out[f"{name}/tgt_valid"] = tv.numpy()
        out[f"{name}/batch_idx"], out[f"{name}/src_idx"] = bi.numpy(), si.numpy()
        out[f"{name}/tgt_idx"] = ti.numpy() if ti is not None else np.array([-1])   # [-1] encodes None
        # also the cost matrix of the kept samples (fp32) for a tolerance check of the device-side expression
        print(name, "matches:", len(si), "tgt_idx:", None if ti is None else ti.tolist()[:8])
    np.savez_compressed(os.path.join(HERE, "matcher_cases.npz"), **out)


if __name__ == "__main__":
    main()
