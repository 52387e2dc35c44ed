"""The small COCO directory shared by make_dataset_golden.py (reference's dataset) and tests/test_sam3_data.py (this
library's): data only."""
import json
import os

import numpy as np

IMAGES = [dict(id=7, file_name="a.png", width=40, height=30), dict(id=3, file_name="b.png", width=64, height=64),
          dict(id=11, file_name="c.png", width=33, height=50)]
CATEGORIES = [dict(id=1, name="Crack"), dict(id=2, name="Pothole")]


def _rle_counts(mask):
    flat = mask.T.reshape(-1)
    counts, cur, run = [], 0, 0
    for v in flat:
        if v == cur:
            run += 1
        else:
            counts.append(run)
            cur, run = v, 1
    counts.append(run)
    return [int(c) for c in counts]


def annotations():
    m = np.zeros((64, 64), np.uint8)
    m[10:20, 5:50] = 1
    m[30:33, 60:64] = 1
    return [
        dict(id=1, image_id=7, category_id=1, bbox=[4.5, 3.5, 16.0, 14.0], segmentation=[[4.5, 3.5, 20.5, 3.5, 20.5, 17.5, 4.5, 17.5]]),
        dict(id=2, image_id=7, category_id=2, bbox=[25.5, 10.5, 10.0, 12.0],
             segmentation=[[25.5, 10.5, 35.5, 10.5, 35.5, 22.5, 25.5, 22.5], [30.5, 1.5, 38.5, 1.5, 38.5, 6.5, 30.5, 6.5]]),
        dict(id=3, image_id=7, category_id=2, bbox=[1.25, 2.75, 7.3, 9.1]),                       # a box without a mask
        dict(id=4, image_id=3, category_id=1, bbox=[5, 10, 59, 23], segmentation=dict(counts=_rle_counts(m), size=[64, 64])),
        dict(id=5, image_id=3, category_id=1, segmentation=[[1.5, 1.5, 9.5, 1.5, 9.5, 9.5, 1.5, 9.5]]),      # no bbox: skipped
    ]


def write_coco_dir(root):
    from PIL import Image as PILImage
    split = os.path.join(root, "train")
    os.makedirs(split, exist_ok=True)
    rng = np.random.default_rng(5)
    for im in IMAGES:
        arr = rng.integers(0, 256, (im["height"], im["width"], 3), dtype=np.uint8)
        PILImage.fromarray(arr).save(os.path.join(split, im["file_name"]))
    with open(os.path.join(split, "_annotations.coco.json"), "w") as f:
        json.dump(dict(images=IMAGES, annotations=annotations(), categories=CATEGORIES), f)
