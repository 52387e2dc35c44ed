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


def _batched_equals_one_by_one(device):
    """launch / collect over several outputs (a decoder's final + auxiliary ones) = the per-output calls, also with images
    without targets (removed from the cost), with a target-index list (more targets than queries) and with the host copy of
    the box counts the collator provides."""
    g = torch.Generator().manual_seed(7)
    for B, Q, counts, L in ((3, 12, [2, 0, 4], 4), (2, 3, [5, 1], 3), (4, 10, [1, 1, 1, 1], 6), (2, 6, [0, 0], 2)):
        T = max(max(counts), 1)
        cxcy = torch.rand(B, T, 2, generator=g) * 0.6 + 0.2
        tgt = torch.cat([cxcy, torch.rand(B, T, 2, generator=g) * 0.3 + 0.05], -1).to(device)
        outs = [{"pred_logits": torch.randn(B, Q, 1, generator=g).to(device),
                 "pred_boxes": torch.cat([torch.rand(B, Q, 2, generator=g) * 0.6 + 0.2,
                                          torch.rand(B, Q, 2, generator=g) * 0.3 + 0.05], -1).to(device)} for _ in range(L)]
        nb = torch.tensor(counts, dtype=torch.long, device=device)
        m = BinaryHungarianMatcherV2(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True)
        for targets in ({"num_boxes": nb, "boxes_padded": tgt},
                        {"num_boxes": nb, "boxes_padded": tgt, "num_boxes_host": tuple(counts)}):
            together = m.collect(m.launch(outs, targets))
            assert len(together) == L
            for o, got in zip(outs, together):
                alone = m(o, {"num_boxes": nb, "boxes_padded": tgt})
                for a, b in zip(got, alone):
                    assert (a is None and b is None) or (a.dtype == torch.int64 and torch.equal(a, b))


def test_batched_matching_equals_one_by_one_cpu():
    _batched_equals_one_by_one("cpu")


@pytest.mark.gpu
def test_batched_matching_equals_one_by_one_gpu():
    _batched_equals_one_by_one("cuda:0")
