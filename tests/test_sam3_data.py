"""Data step before the path (rows a14 / f-4): synthetic samples, rank sharding with DistributedSampler semantics,
the COCO dataset restatement (polygon / RLE decoding, box normalisation, query text), the shipped YAMLs."""
import glob
import json
import os

import numpy as np
import pytest
import torch
import yaml

from sam3_lora_amd import sam3_data as SD
from sam3_lora_amd import trainer as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_sample_layout_and_determinism():
    a, b = SD.synthetic_datapoint(3, resolution=112, source=128), SD.synthetic_datapoint(3, resolution=112, source=128)
    img = a.images[0].data
    assert img.shape == (3, 112, 112) and img.dtype == torch.float32 and -1.0 <= img.min() and img.max() <= 1.0
    assert torch.equal(img, b.images[0].data)
    assert not torch.equal(img, SD.synthetic_datapoint(4, resolution=112, source=128).images[0].data)
    assert len(a.images[0].objects) == 2 and a.find_queries[0].query_text == "crack" and a.find_queries[0].is_exhaustive
    for o in a.images[0].objects:
        x1, y1, x2, y2 = o.bbox.tolist()              # normalised xyxy (the reference's storage quirk)
        assert 0 <= x1 < x2 <= 1 and 0 <= y1 < y2 <= 1
        assert o.segment.dtype == torch.bool and o.segment.shape == (112, 112)
        ys, xs = torch.nonzero(o.segment, as_tuple=True)
        assert abs(xs.min().item() - round(x1 * 112)) <= 1 and abs(ys.max().item() + 1 - round(y2 * 112)) <= 1
    batch = SD.collate_fn_api([a, SD.synthetic_datapoint(4, 112, 128)], dict_key="input", with_seg_masks=True)["input"]
    assert batch.img_batch.shape == (2, 3, 112, 112) and batch.find_text_batch == ["crack"]
    t = batch.find_targets[0]
    assert t.num_boxes.tolist() == [2, 2] and t.boxes.shape == (4, 4) and t.boxes_padded.shape == (2, 2, 4)
    assert t.segments.shape == (4, 112, 112) and t.is_valid_segment.all()
    assert batch.find_inputs[0].input_boxes.shape == (0, 2, 4) and batch.find_inputs[0].input_boxes_mask.shape == (2, 0)


@pytest.mark.parametrize("n,world", [(10, 2), (11, 2), (64, 8), (7, 4), (5, 8)])
def test_shards_are_distributed_sampler_shards(n, world):
    from torch.utils.data.distributed import DistributedSampler
    for epoch in (0, 3):
        shards = [SD.shard_indices(n, r, world, epoch=epoch, shuffle=True, seed=5) for r in range(world)]
        assert len({len(s) for s in shards}) == 1                       # equally long
        assert set(i for s in shards for i in s) == set(range(n))       # union = dataset
        for r in range(world):                                          # and exactly torch's sampler
            ds = DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=True, seed=5)
            ds.set_epoch(epoch)
            assert list(ds) == shards[r]
    assert SD.shard_indices(n, 0, 1, shuffle=False) == list(range(n))
    assert SD.shard_indices(n, 0, world, epoch=0, seed=5) != SD.shard_indices(n, 0, world, epoch=1, seed=5) or n < 3


def test_default_data_builder_shards_by_rank(monkeypatch):
    cfg = {"training": {"data_dir": "synthetic:6", "batch_size": 2}}
    seen = []
    for rank in range(2):
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "2")
        loader = T.default_data_builder(cfg, "train")
        assert len(loader) == 2
        seen.append(loader.indices())
    assert sorted(seen[0] + seen[1]) == list(range(6))
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    val = T.default_data_builder(cfg, "valid")
    assert len(val.dataset) == 1 and val.indices() == [0]


def test_rle_and_polygon_decoding():
    # run lengths, column-major, first run = zeros
    m = SD.rle_decode([2, 3, 4, 3], 3, 4)
    assert m.int().t().flatten().tolist() == [0, 0, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1]
    # compressed string: encode with the inverse of the published varint scheme, decode back
    def enc(counts):
        s = ""
        for i, c in enumerate(counts):
            x = c - counts[i - 2] if i > 2 else c
            more = True
            while more:
                ch = x & 0x1F
                x >>= 5
                more = not (x == -1 and (ch & 0x10)) if (ch & 0x10) else x != 0
                if more:
                    ch |= 0x20
                s += chr(ch + 48)
        return s
    for counts in ([5, 3, 10, 2], [0, 7, 1000, 3, 2, 40, 9], [70000, 3, 2, 1, 100, 5]):
        assert SD.rle_counts_from_string(enc(counts)) == counts
    # axis-aligned polygons rasterise to exact pixel rectangles; union of two polygons
    rect = SD.polygon_to_mask([2, 3, 10, 3, 10, 8, 2, 8], 12, 16)
    want = torch.zeros(12, 16, dtype=torch.bool)
    want[3:8, 2:10] = True
    assert torch.equal(rect, want)
    assert torch.equal(SD.polygon_to_mask([10, 8, 2, 8, 2, 3, 10, 3], 12, 16), want)        # winding / start vertex
    both = SD.segmentation_to_mask([[2, 3, 10, 3, 10, 8, 2, 8], [12, 0, 15, 0, 15, 2, 12, 2]], 12, 16)
    assert both.sum().item() == 40 + 6
    tri = SD.polygon_to_mask([0, 0, 8, 0, 0, 8], 8, 8)
    assert abs(tri.sum().item() - 32) <= 5 and tri[0, 0] and not tri[7, 7]
    assert torch.equal(tri, tri.t())                                                        # symmetric shape
    assert torch.equal(SD.segmentation_to_mask({"size": [3, 4], "counts": [2, 3, 4, 3]}, 3, 4), m)


def test_coco_dataset_sample(tmp_path):
    from PIL import Image as PILImage
    split = tmp_path / "train"
    split.mkdir()
    rng = np.random.default_rng(0)
    PILImage.fromarray(rng.integers(0, 255, (60, 80, 3), dtype=np.uint8)).save(split / "a.png")
    PILImage.fromarray(rng.integers(0, 255, (60, 80, 3), dtype=np.uint8)).save(split / "b.png")
    coco = {"images": [{"id": 7, "file_name": "a.png", "width": 80, "height": 60},
                       {"id": 3, "file_name": "b.png", "width": 80, "height": 60}],
            "categories": [{"id": 1, "name": "Crack"}, {"id": 2, "name": "Spall"}],
            "annotations": [
                {"id": 1, "image_id": 7, "category_id": 1, "bbox": [8, 6, 40, 30],
                 "segmentation": [[8, 6, 48, 6, 48, 36, 8, 36]]},
                {"id": 2, "image_id": 7, "category_id": 1, "bbox": [0, 0, 16, 12],
                 "segmentation": {"size": [60, 80], "counts": [0] + [12, 48] * 15 + [12, 48 + 64 * 60]}},
                {"id": 3, "image_id": 7, "category_id": 2, "bbox": [60, 40, 10, 10]}]}
    (split / "_annotations.coco.json").write_text(json.dumps(coco))
    ds = SD.COCOSegmentDataset(str(tmp_path), "train", resolution=112)
    assert len(ds) == 2 and ds.image_ids == [3, 7]
    empty, dp = ds[0], ds[1]
    assert empty.find_queries[0].query_text == "object" and empty.images[0].objects == []
    assert dp.find_queries[0].query_text == "crack"                 # most common category, lower-cased
    img = dp.images[0].data
    assert img.shape == (3, 112, 112) and img.min() >= -1 and img.max() <= 1
    o0, o1, o2 = dp.images[0].objects
    assert torch.allclose(o0.bbox, torch.tensor([0.1, 0.1, 0.6, 0.6]))          # normalised xyxy
    assert o0.segment.shape == (112, 112) and abs(o0.segment.float().mean().item() - 0.25) < 0.01
    assert o1.segment[:22, :22].all() and not o1.segment[23:, :].any()
    assert o2.segment is None
    batch = SD.collate_fn_api([dp, empty], dict_key="input", with_seg_masks=True)["input"]
    assert batch.find_targets[0].num_boxes.tolist() == [3, 0]
    assert batch.find_targets[0].is_valid_segment.tolist() == [True, True, False]
    with pytest.raises(FileNotFoundError):
        SD.COCOSegmentDataset(str(tmp_path), "valid")


def test_shipped_yamls_have_every_consumed_key():
    files = sorted(glob.glob(os.path.join(ROOT, "configs", "*.yaml")))
    assert os.path.join(ROOT, T.DEFAULT_CONFIG) in files
    for f in files:
        cfg = yaml.safe_load(open(f))
        T.lora_config_from(cfg)
        for k in ("data_dir", "batch_size", "learning_rate", "weight_decay", "num_epochs"):
            assert k in cfg["training"], (f, k)
        assert "output_dir" in cfg["output"]
    full = yaml.safe_load(open(os.path.join(ROOT, "configs", "full_lora_config.yaml")))["lora"]
    assert (full["rank"], full["alpha"], full["dropout"]) == (32, 64, 0.1)      # the reference's literal values


def test_numpy_rasteriser_is_the_scalar_rule_bit_for_bit():
    """polygon_to_mask (numpy: all edges' run boundaries at once + column-major parity fill) == polygon_to_mask_loops (the
    scalar restatement of pycocotools' rleFrPoly) on random polygons incl. vertices outside the image, integer / half-integer
    vertices and degenerate edges."""
    import random
    random.seed(1)
    for trial in range(150):
        h, w = random.choice([17, 64, 100, 333]), random.choice([23, 64, 128, 257])
        k = random.randint(3, 12)
        if trial % 3:
            poly = [v for _ in range(k) for v in (random.uniform(-10, w + 10), random.uniform(-10, h + 10))]
        else:
            poly = [v for _ in range(k) for v in (random.randint(0, w) + random.choice([0, 0.5]), random.randint(0, h) + random.choice([0, 0.5]))]
        if trial % 7 == 0:
            poly += poly[-2:]                       # a zero-length edge
        assert torch.equal(SD.polygon_to_mask(poly, h, w), SD.polygon_to_mask_loops(poly, h, w)), (trial, poly)


def test_coco_dataset_matches_the_references_getitem(tmp_path):
    """tests/golden/dataset_cases.npz: the reference's own COCOSegmentDataset.__getitem__ on the small COCO directory of
    dataset_case_defs.py (make_dataset_golden.py) -- image tensor (digest of the float bits + a strided sample), normalised
    boxes and areas BIT for bit (the fp32 order of operations: scale, then divide by the resolution), nearest-resized masks,
    query text, original size, object ids; samples through the worker pool are the same objects."""
    import hashlib
    import dataset_case_defs as C
    g = np.load(os.path.join(ROOT, "tests", "golden", "dataset_cases.npz"))
    C.write_coco_dir(str(tmp_path))
    ds = SD.COCOSegmentDataset(str(tmp_path), split="train")
    assert len(ds) == int(g["n"])
    for i in range(len(ds)):
        dp = ds[i]
        img, q = dp.images[0], dp.find_queries[0]
        a = img.data.numpy()
        assert a.shape == tuple(g[f"{i}/image_shape"]) and a.dtype == np.float32
        assert np.array_equal(a[:, ::37, ::41], g[f"{i}/image_sample"])
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g[f"{i}/image_sha256"])
        assert tuple(img.size) == tuple(g[f"{i}/size"]) and q.query_text == str(g[f"{i}/text"])
        assert list(q.object_ids_output) == g[f"{i}/object_ids_output"].tolist()
        assert tuple(q.inference_metadata.original_size) == tuple(g[f"{i}/original_size"])
        assert q.inference_metadata.coco_image_id == int(g[f"{i}/coco_image_id"])
        assert len(img.objects) == int(g[f"{i}/n_objects"])
        for j, o in enumerate(img.objects):
            assert o.bbox.numpy().tobytes() == g[f"{i}/obj{j}/bbox"].tobytes(), (i, j, o.bbox, g[f"{i}/obj{j}/bbox"])
            assert np.float32(o.area).tobytes() == np.float32(g[f"{i}/obj{j}/area"]).tobytes()
            assert o.object_id == int(g[f"{i}/obj{j}/object_id"])
            key = f"{i}/obj{j}/segment"
            assert (o.segment is None) == (key not in g.files)
            if o.segment is not None:
                assert np.array_equal(np.packbits(o.segment.numpy()), g[key])
    col = lambda s: SD.collate_fn_api(s, dict_key="input", with_seg_masks=True)
    plain = list(SD.ShardedLoader(ds, 2, col, shuffle=True, seed=1))
    pooled = list(SD.ShardedLoader(ds, 2, col, shuffle=True, seed=1, num_workers=3, prefetch=2))
    assert len(plain) == len(pooled) == 2
    for x, y in zip(plain, pooled):
        assert torch.equal(x["input"].img_batch, y["input"].img_batch) and x["input"].find_text_batch == y["input"].find_text_batch
        assert torch.equal(x["input"].find_targets[0].boxes, y["input"].find_targets[0].boxes)
        assert torch.equal(x["input"].find_targets[0].segments, y["input"].find_targets[0].segments)
