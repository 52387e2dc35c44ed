"""Consumers of the adapter files (row f-3): post-processing on the CPU; on the GPU the whole chain train -> save ->
infer / validate / merge at the tiny model configuration."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from sam3_lora_amd import inference as I


def test_mask_nms_and_merge():
    m = torch.zeros(4, 16, 16)
    m[0, 2:8, 2:8] = 1
    m[1, 3:9, 3:9] = 1          # IoU with 0: 25/47 = 0.53
    m[2, 10:14, 10:14] = 1
    m[3, 2:8, 2:8] = 1          # duplicate of 0 with a low score
    p = torch.tensor([0.9, 0.8, 0.7, 0.2])
    assert I.nms_masks(p, m, 0.3, 0.5).tolist() == [True, False, True, False]
    assert I.nms_masks(p, m, 0.3, 0.6).tolist() == [True, True, True, False]
    assert I.nms_masks(p, m, 0.95, 0.5).tolist() == [False] * 4
    masks, scores, boxes = I.apply_sam3_nms(torch.logit(p)[:, None], (m - 0.5) * 10, torch.rand(4, 4), 0.3, 0.5, max_detections=1)
    assert masks.shape == (1, 16, 16) and abs(scores.item() - 0.9) < 1e-6
    mm, ms, mb = I.merge_overlapping_masks(m.bool(), p, torch.arange(16.).view(4, 4), 0.15)
    assert len(mm) == 2 and mm[0].sum().item() == 47 and ms.tolist() == pytest.approx([0.9, 0.7])


def test_average_precision_known_answers():
    gt = {0: [np.pad(np.ones((4, 4), bool), ((0, 4), (0, 4)))], 1: [np.pad(np.ones((4, 4), bool), ((4, 0), (4, 0)))]}
    perfect = [(0, 0.9, gt[0][0]), (1, 0.8, gt[1][0])]
    r = I.mask_average_precision(perfect, gt)
    assert r["mAP"] == pytest.approx(1.0) and r["f1_50"] == pytest.approx(1.0)
    half = gt[0][0].copy()
    half[2:4] = False                           # IoU 0.5 with the ground truth: counts at threshold 0.5 only
    r = I.mask_average_precision([(0, 0.9, half), (1, 0.8, gt[1][0])], gt)
    # at IoU > 0.5 the first-ranked detection is a false positive: precision 1/2 up to recall 1/2, nothing beyond ->
    # 0.5 * 51/101 of the 101 recall points
    assert r["mAP50"] == pytest.approx(1.0) and r["mAP75"] == pytest.approx(0.5 * 51 / 101, abs=1e-6)
    assert r["mAP"] == pytest.approx((1.0 + 9 * 0.5 * 51 / 101) / 10, abs=1e-6)
    # a false positive ranked first halves the precision envelope at low recall
    r = I.mask_average_precision([(0, 0.95, ~gt[0][0])] + perfect, gt)
    assert r["mAP50"] == pytest.approx(2 / 3, abs=0.01) and r["precision50"] == pytest.approx(2 / 3)
    assert I.mask_average_precision([], gt)["mAP"] == 0.0


@pytest.mark.gpu
def test_train_save_infer_validate_merge_chain(tmp_path, monkeypatch):
    """Adapters trained by the CLI's trainer are consumed by inference / validation, and the merged (adapter-free) model
    gives the same outputs."""
    import sam3_lora_amd.sam3_image as SI
    from PIL import Image as PILImage
    from sam3_lora_amd import trainer as T
    from sam3_lora_amd.sam3_data import SyntheticSegmentDataset
    tiny = dict(SI.TINY_CONFIG, text=dict(SI.TINY_CONFIG["text"], vocab_size=49408, context_length=32))
    real_build = SI.build_sam3_image_model
    monkeypatch.setattr(SI, "build_sam3_image_model", lambda **kw: real_build(**dict(kw, config=tiny)))
    import sam3_lora_amd.sam3_data as SD
    monkeypatch.setattr(SD, "SyntheticSegmentDataset",
                        lambda n, split="train", **kw: SyntheticSegmentDataset(n, split=split, resolution=112, source=128))
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs",
                                           "minimal_lora_config.yaml")))
    cfg["training"].update(num_epochs=2, data_dir="synthetic:4", learning_rate=1e-3)
    cfg["output"]["output_dir"] = str(tmp_path / "out")
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    tr = T.SAM3TrainerNative(str(path))
    res = tr.train()
    assert len(res["history"]) == 2 and (tmp_path / "out" / "best_lora_weights.pt").exists()
    img = tmp_path / "img.png"
    PILImage.fromarray(np.random.default_rng(0).integers(0, 255, (90, 120, 3), dtype=np.uint8)).save(img)
    inf = I.SAM3LoRAInference(str(path), resolution=112, detection_threshold=0.0)
    out = inf.predict(str(img), ["crack", "object"])
    assert out[0]["num_detections"] == tiny["num_queries"] and out[0]["masks"].shape == (tiny["num_queries"], 90, 120)
    assert out[0]["boxes"].shape == (tiny["num_queries"], 4) and set(out) == {0, 1, "_image"}
    assert inf.visualize(out, str(tmp_path / "vis.png")) == 2 * tiny["num_queries"] and (tmp_path / "vis.png").exists()
    merged = I.SAM3LoRAInference(str(path), resolution=112, detection_threshold=0.0, merge=True)
    assert not any(isinstance(m, I.LoRALinear) for m in merged.model.modules())
    out_m = merged.predict(str(img), ["crack"])
    assert np.allclose(out_m[0]["scores"], out[0]["scores"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out_m[0]["boxes"], out[0]["boxes"], rtol=1e-3, atol=1e-2)
    metrics = I.validate(str(path), str(tmp_path / "out" / "best_lora_weights.pt"), "unused", prob_threshold=0.0,
                         dataset=SyntheticSegmentDataset(3, split="valid", resolution=112, source=128))
    assert metrics["images"] == 3 and metrics["num_ground_truth"] == 6 and 0.0 <= metrics["mAP"] <= 1.0


# ---------------------------------------------------------------------------------------------------------------------
# The validation script's post-processing against the REFERENCE's own functions (tests/golden/consumer_cases.npz, written by
# make_consumer_golden.py: validate_sam3_lora.py:232-352 executed unmodified + sam3/perflib/nms.py): the surviving set and its
# order, scores, boxes and masks are bit-exact (integer / index work + the same sigmoid).
def test_nms_and_overlap_merge_match_the_reference_functions():
    import os
    import numpy as np
    import torch
    import consumer_case_defs as C
    from sam3_lora_amd import inference as I
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consumer_cases.npz"))
    for name, n, side, clusters, seed, pt, it, k, mt in C.CASES:
        logits, masks, boxes = C.make_case(n, side, clusters, seed)
        fm, fs, fb = I.apply_sam3_nms(logits, masks, boxes, prob_threshold=pt, nms_iou_threshold=it, max_detections=k)
        assert np.array_equal(fs.numpy(), gold[f"{name}/nms_scores"]), name
        assert np.array_equal(fb.numpy(), gold[f"{name}/nms_boxes"]), name
        assert np.array_equal(fm.numpy(), gold[f"{name}/nms_masks"]), name
        if len(fm) > 0:
            mm, ms, mb = I.merge_overlapping_masks(fm > 0.5, fs, fb, iou_threshold=mt)
        else:
            mm, ms, mb = fm > 0.5, fs, fb
        assert len(mm) == int(gold[f"{name}/merged_count"]), name
        assert np.array_equal(np.packbits(mm.numpy().astype(bool), axis=-1), gold[f"{name}/merged_masks"]), name
        assert np.array_equal(ms.numpy(), gold[f"{name}/merged_scores"]) and np.array_equal(mb.numpy(), gold[f"{name}/merged_boxes"]), name
