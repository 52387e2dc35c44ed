#!/usr/bin/env python3
"""
Golden vectors for the data step before the path: the REFERENCE's own ``COCOSegmentDataset.__getitem__``
(``/root/reference/train_sam3_lora_native.py:95-232``) run in the build container on a small COCO directory that this script
writes (PNG images of seeded noise; axis-aligned rectangular polygons with half-integer vertices, one multi-polygon object,
one uncompressed-RLE object, one image without annotations, two categories) -> ``dataset_cases.npz``.

Third-party pieces the image lacks are stood in for by their PUBLISHED semantics, each a few lines:
  * ``torchvision.transforms.v2``: ``ToImage`` (PIL -> uint8 CHW tensor), ``ToDtype(float32, scale=True)`` (/ 255),
    ``Normalize(mean, std)`` ((x - mean) / std per channel), ``Compose``;
  * ``pycocotools.mask``: ``frPyObjects`` / ``merge`` / ``decode`` evaluated ANALYTICALLY for the shapes this fixture uses
    -- an axis-aligned rectangle with half-integer corner coordinates covers exactly the pixels whose integer coordinates
    lie strictly inside it (no pixel centre is on an edge, so every correct rasteriser agrees), RLE counts are run lengths in
    column-major order.  This stand-in knows nothing of this library's rasteriser.
What the fixture pins bit for bit: the image tensor (PIL bilinear resize, /255, (x - .5) / .5), the fp32 arithmetic ORDER of the
normalised boxes and areas, the nearest-neighbour resize of the masks, the query text rule, ``original_size`` and the
object ids.  What it cannot pin is pycocotools' rasterisation of general polygons (sam3_data.polygon_to_mask_loops restates
the published rule; pycocotools is absent here): stated as "parity unpinned" in DESIGN.md.
"""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)

import numpy as np
import torch
from PIL import Image as PILImage

import dataset_case_defs as C
import sam3_manifest


def install_stand_ins():
    sam3_manifest._install_stubs()
    v2 = types.ModuleType("torchvision.transforms.v2")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToImage:
        def __call__(self, pil):
            return torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1)

    class ToDtype:
        def __init__(self, dtype, scale=False):
            self.dtype, self.scale = dtype, scale

        def __call__(self, x):
            return x.to(self.dtype) / 255.0 if self.scale else x.to(self.dtype)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std
    v2.Compose, v2.ToImage, v2.ToDtype, v2.Normalize = Compose, ToImage, ToDtype, Normalize
    sys.modules["torchvision.transforms.v2"] = v2
    tv_t = sam3_manifest._StubModule("torchvision.transforms")       # permissive for every other name the model files import
    tv_t.v2 = v2
    sys.modules["torchvision.transforms"] = tv_t

    m = types.ModuleType("pycocotools.mask")

    def rect(poly, h, w):
        xs, ys = poly[0::2], poly[1::2]
        assert len(poly) == 8 and len(set(xs)) == 2 and len(set(ys)) == 2 and all(abs(v * 2 % 2 - 1) < 1e-9 for v in poly), \
            "the analytic stand-in only knows half-integer axis-aligned rectangles"
        yy, xx = np.mgrid[0:h, 0:w]
        return ((xx > min(xs)) & (xx < max(xs)) & (yy > min(ys)) & (yy < max(ys))).astype(np.uint8)
    m.frPyObjects = lambda polys, h, w: [rect(p, h, w) for p in polys]
    m.merge = lambda masks: np.clip(np.sum(masks, axis=0), 0, 1).astype(np.uint8)

    def decode(obj):
        if isinstance(obj, dict):       # uncompressed RLE: run lengths, column-major, starting with zeros
            h, w = obj["size"]
            vals = np.repeat(np.arange(len(obj["counts"])) % 2, obj["counts"]).astype(np.uint8)
            return vals.reshape(w, h).T.copy()
        return np.asarray(obj, np.uint8)
    m.decode = decode
    sys.modules["pycocotools.mask"] = m
    pk = types.ModuleType("pycocotools")
    pk.mask = m
    sys.modules["pycocotools"] = pk


def main():
    install_stand_ins()
    sam3_reg = types.ModuleType("sam3")
    sam3_reg.__path__ = [os.path.join(REF, "sam3")]
    sys.modules.setdefault("sam3", sam3_reg)
    # the script also imports the model builder (video predictors, io utilities ...), which the dataset never touches:
    # those module names resolve to permissive stand-ins; the record types the dataset RETURNS are the reference's own
    for name in ("sam3.model_builder", "sam3.train.masks_ops", "sam3.train.loss.loss_fns", "sam3.train.loss.sam3_loss",
                 "sam3.train.matcher", "tqdm"):
        sys.modules[name] = sam3_manifest._StubModule(name)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_train_native", os.path.join(REF, "train_sam3_lora_native.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.__file__.startswith(REF)
    res = {}
    with tempfile.TemporaryDirectory() as d:
        C.write_coco_dir(d)
        ds = mod.COCOSegmentDataset(d, split="train")
        res["n"] = np.int64(len(ds))
        for i in range(len(ds)):
            dp = ds[i]
            img = dp.images[0]
            a = img.data.numpy()                        # 12 MB of floats per sample: pinned by digest + a strided sample
            import hashlib
            res[f"{i}/image_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest())
            res[f"{i}/image_shape"] = np.array(a.shape)
            res[f"{i}/image_sample"] = a[:, ::37, ::41].copy()
            res[f"{i}/size"] = np.array(img.size)
            q = dp.find_queries[0]
            res[f"{i}/text"] = np.array(q.query_text)
            res[f"{i}/object_ids_output"] = np.array(q.object_ids_output, np.int64)
            res[f"{i}/original_size"] = np.array(q.inference_metadata.original_size)
            res[f"{i}/coco_image_id"] = np.int64(q.inference_metadata.coco_image_id)
            res[f"{i}/n_objects"] = np.int64(len(img.objects))
            for j, o in enumerate(img.objects):
                res[f"{i}/obj{j}/bbox"] = o.bbox.numpy()
                res[f"{i}/obj{j}/area"] = np.float32(o.area)
                res[f"{i}/obj{j}/object_id"] = np.int64(o.object_id)
                if o.segment is not None:
                    res[f"{i}/obj{j}/segment"] = np.packbits(o.segment.numpy())
    out = os.path.join(HERE, "dataset_cases.npz")
    np.savez_compressed(out, **res)
    print("samples:", int(res["n"]), "arrays:", len(res), "bytes:", os.path.getsize(out))


if __name__ == "__main__":
    main()
