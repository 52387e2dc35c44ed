"""Hungarian matcher (sam3_lora_amd/matcher.py) against indices returned by the reference's
BinaryHungarianMatcherV2 on the same seeded inputs (tests/golden/matcher_cases.npz) -- bit-exact int64."""
import os

import numpy as np
import pytest
import torch

from matcher_case_defs import CASES
from sam3_lora_amd.matcher import BinaryHungarianMatcherV2, box_cxcywh_to_xyxy, generalized_box_iou


def _run(golden_dir, name, device):
    g = np.load(os.path.join(golden_dir, "matcher_cases.npz"))
    kw, B, Q, nb, repeats, rb, uo, ut = CASES[name]
    t = lambda k: torch.from_numpy(g[f"{name}/{k}"]).to(device)
    m = BinaryHungarianMatcherV2(**kw)
    bi, si, ti = m({"pred_logits": t("logits"), "pred_boxes": t("boxes")},
                   {"num_boxes": t("num_boxes"), "boxes_padded": t("tgt")}, repeats=repeats, repeat_batch=rb,
                   out_is_valid=t("out_valid") if uo else None, target_is_valid_padded=t("tgt_valid") if ut else None)
    assert bi.dtype == si.dtype == torch.int64
    assert np.array_equal(bi.cpu().numpy(), g[f"{name}/batch_idx"]), name
    assert np.array_equal(si.cpu().numpy(), g[f"{name}/src_idx"]), name
    want_t = g[f"{name}/tgt_idx"]
    if want_t.tolist() == [-1]:
        assert ti is None
    else:
        assert ti is not None and np.array_equal(ti.cpu().numpy(), want_t), name


@pytest.mark.parametrize("name", sorted(CASES))
def test_indices_bit_exact_cpu(golden_dir, name):
    _run(golden_dir, name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_indices_bit_exact_gpu(golden_dir, name):
    """cost matrix built on the MI355X in fp32, assignment on the host: same indices as the CPU reference."""
    _run(golden_dir, name, "cuda:0")


def test_giou_properties():
    b = torch.tensor([[0.5, 0.5, 0.2, 0.2], [0.3, 0.3, 0.1, 0.4]])
    x = box_cxcywh_to_xyxy(b)
    g = generalized_box_iou(x, x)
    assert torch.allclose(torch.diag(g), torch.ones(2), atol=1e-6)
    assert g[0, 1] < 0 and torch.allclose(g, g.t())
