"""
The SAM3 image model around the adapted trunk (row a14) against the end-to-end fixture produced by the reference's own
classes at a tiny configuration (tests/golden/make_e2e_golden.py -> e2e_tiny.npz): state-dict keys, collated batch,
eval / training forward, matcher indices; and -- on the GPU, through the HIP adapter path -- the loss dictionary, A/B
gradients, A/B after AdamW and the loss curve of the native training loop.
"""
import os

import numpy as np
import pytest
import torch

import e2e_case_defs as D
from loss_case_defs import CLI_LOSS_CFG

from sam3_lora_amd.sam3_data import (Datapoint, FindQueryLoaded, Image, InferenceMetadata, Object, collate_fn_api)
from sam3_lora_amd.sam3_image import SAM3Output, TINY_CONFIG, build_sam3_image_model

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_tiny.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def state_dict_of(gold):
    sd = {}
    for k in gold.files:
        if not k.startswith("sd/"):
            continue
        name = k[3:]
        if name.endswith(".re"):
            sd[name[:-3]] = torch.complex(torch.from_numpy(gold[k]), torch.from_numpy(gold["sd/" + name[:-3] + ".im"]))
        elif not name.endswith(".im"):
            sd[name] = torch.from_numpy(gold[k])
    return sd


def build(gold, device="cpu", **kw):
    assert TINY_CONFIG == D.TINY
    model = build_sam3_image_model(device=device, eval_mode=False, config=TINY_CONFIG, tokenizer=D.toy_tokenizer, **kw)
    model.load_state_dict({k: v.to(device) for k, v in state_dict_of(gold).items()}, strict=True)
    return model


def make_batch():
    dps = []
    for i, ((text, boxes), img) in enumerate(zip(D.SAMPLES, D.make_images())):
        objs = [Object(bbox=torch.tensor(b, dtype=torch.float32), area=b[2] * b[3], object_id=j, segment=D.box_mask(b))
                for j, b in enumerate(boxes)]
        q = FindQueryLoaded(query_text=text, image_id=0, object_ids_output=list(range(len(objs))), is_exhaustive=True,
                            query_processing_order=0,
                            inference_metadata=InferenceMetadata(coco_image_id=i, original_image_id=i,
                                                                 original_category_id=0, original_size=(D.RES, D.RES),
                                                                 object_id=-1, frame_index=-1))
        dps.append(Datapoint(find_queries=[q], images=[Image(data=img, objects=objs, size=(D.RES, D.RES))]))
    return collate_fn_api(dps, dict_key="input", with_seg_masks=True)["input"]


def close(a, ref, rtol, what):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    scale = max(float(np.abs(ref).max()), 1e-6)
    err = float(np.abs(a - ref).max()) / scale
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    assert err <= rtol, f"{what}: max-abs error / max|ref| = {err:.3e} > {rtol}"


def check_outputs(gold, tag, out, rtol, indices=True):
    n = 0
    for k in gold.files:
        if not k.startswith(tag + "/"):
            continue
        parts = k.split("/")[1:]
        node = out
        if parts[0].startswith("aux"):
            node = out["aux_outputs"][int(parts[0][3:])]
            parts = parts[1:]
        if parts[0] == "indices":
            if indices:
                got = torch.stack([node["indices"][0], node["indices"][1]]).cpu().numpy()
                assert np.array_equal(got, gold[k]), f"{k}: matcher indices differ\n{got}\n{gold[k]}"
                n += 1
            continue
        close(node[parts[0]], gold[k], rtol, k)
        n += 1
    return n


def test_state_dict_keys_are_the_references(gold):
    model = build_sam3_image_model(device="cpu", eval_mode=False, config=TINY_CONFIG, tokenizer=D.toy_tokenizer)
    ours = list(model.state_dict().keys())
    theirs = [str(k) for k in gold["sd_keys"]]
    assert sorted(ours) == sorted(theirs)
    sd = state_dict_of(gold)
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k


def test_collated_batch_matches_reference(gold):
    b = make_batch()
    assert list(b.find_text_batch) == [str(t) for t in gold["batch/texts"]]
    assert torch.equal(b.img_batch, torch.from_numpy(gold["batch/img_batch"]))
    for k in gold.files:
        if k.startswith("batch/find_input/") or k.startswith("batch/find_target/"):
            _, grp, name = k.split("/")
            got = getattr(b.find_inputs[0] if grp == "find_input" else b.find_targets[0], name)
            ref = gold[k]
            assert tuple(got.shape) == ref.shape, (k, got.shape, ref.shape)
            assert str(got.numpy().dtype) == str(ref.dtype), (k, got.dtype, ref.dtype)
            assert np.array_equal(got.numpy(), ref), k


def test_eval_and_training_forward_match_reference_cpu(gold):
    model = build(gold)
    batch = make_batch()
    model.eval()
    with torch.no_grad():
        out = model(batch)[0]
    assert isinstance(model(batch), SAM3Output)
    assert check_outputs(gold, "eval", out, 2e-5) >= 6
    model.train()
    out = model(batch)[0]
    assert len(out["aux_outputs"]) == TINY_CONFIG["dec_layers"] - 1
    assert check_outputs(gold, "train", out, 2e-5) >= 20
    # without per-layer checkpointing and without the in-forward matching: same tensors
    model2 = build(gold, act_checkpoint=False, match_in_forward=False)
    model2.train()
    out2 = model2(batch)[0]
    assert "indices" not in out2
    check_outputs(gold, "train", out2, 2e-5, indices=False)
