"""
Batch layout of the SAM3 image-training step and the synthetic benchmark data (SURVEY section 2 "data layout",
section 8(d) "synthetic inputs").

Restates, for rows a14 / f-4:
  * the per-sample records of ``sam3/train/data/sam3_image_dataset.py:30-132`` (``InferenceMetadata``, ``FindQueryLoaded``,
    ``Object``, ``Image``, ``Datapoint``);
  * the batched records of ``sam3/model/data_misc.py:46-166`` (``FindStage``, ``BatchedFindTarget``,
    ``BatchedInferenceMetadata``, ``BatchedDatapoint``) -- same field names, dtypes and stacking axes;
  * ``collate_fn_api`` (``sam3/train/data/collator.py:136-360``) for the image case: one stage per
    ``query_processing_order``, texts de-duplicated in first-seen order, packed + padded target boxes, box / point
    prompts padded to the longest with True in the masks;
  * the synthetic sample of SURVEY section 8(d): a seeded uniform-noise RGB image resized and normalised as
    ``COCOSegmentDataset.__getitem__`` does (``train_sam3_lora_native.py:101-108``: bilinear resize to 1008 x 1008,
    scale to [0, 1], mean = std = 0.5), two rectangular objects stored as normalised xyxy (the reference's box-format
    quirk, ``:131-142``) with matching boolean masks, query text "crack".

Pure host-side Python / torch; nothing here touches the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

__all__ = ["InferenceMetadata", "FindQueryLoaded", "Object", "Image", "Datapoint", "FindStage", "BatchedFindTarget",
           "BatchedInferenceMetadata", "BatchedDatapoint", "collate_fn_api", "SyntheticSegmentDataset",
           "synthetic_datapoint", "shard_indices", "ShardedLoader", "COCOSegmentDataset", "segmentation_to_mask",
           "polygon_to_mask", "polygon_to_mask_loops", "polygon_marks", "rle_decode", "rle_counts_from_string"]


# ------------------------------------------------------------------------------------------ per-sample records --
@dataclass
class InferenceMetadata:
    coco_image_id: int
    original_image_id: int
    original_category_id: int
    original_size: Tuple[int, int]
    object_id: int
    frame_index: int
    is_conditioning_only: Optional[bool] = False


@dataclass
class FindQueryLoaded:
    query_text: str
    image_id: int
    object_ids_output: List[int]
    is_exhaustive: bool
    query_processing_order: int = 0
    input_bbox: Optional[torch.Tensor] = None
    input_bbox_label: Optional[torch.Tensor] = None
    input_points: Optional[torch.Tensor] = None
    semantic_target: Optional[torch.Tensor] = None
    is_pixel_exhaustive: Optional[bool] = None
    inference_metadata: Optional[InferenceMetadata] = None


@dataclass
class Object:
    bbox: torch.Tensor
    area: float
    object_id: Optional[int] = -1
    frame_index: Optional[int] = -1
    segment: Optional[Union[torch.Tensor, dict]] = None
    is_crowd: bool = False
    source: Optional[str] = None


@dataclass
class Image:
    data: Any
    objects: List[Object]
    size: Tuple[int, int]
    blurring_mask: Optional[Dict[str, Any]] = None


@dataclass
class Datapoint:
    find_queries: List[FindQueryLoaded]
    images: List[Image]
    raw_images: Optional[List[Any]] = None


# --------------------------------------------------------------------------------------------- batched records --
# (field -> (dtype, stack axis)); list-valued fields not named here stay Python lists
_FIND_STAGE = dict(img_ids=(torch.long, 0), text_ids=(torch.long, 0), input_boxes=(torch.float, 1),
                   input_boxes_mask=(torch.bool, 0), input_boxes_label=(torch.long, 1), input_points=(torch.float, 0),
                   input_points_mask=(torch.bool, 0))
_FIND_TARGET = dict(num_boxes=(torch.long, 0), boxes=(torch.float, 0), boxes_padded=(torch.float, 0),
                    repeated_boxes=(torch.float, 0), segments=(torch.bool, 0), semantic_segments=(torch.bool, 0),
                    is_valid_segment=(torch.bool, 0), is_exhaustive=(torch.bool, 0), object_ids=(torch.long, 0),
                    object_ids_padded=(torch.long, 0))
_METADATA = dict(coco_image_id=(torch.long, 0), original_image_id=(torch.long, 0), original_category_id=(torch.int, 0),
                 original_size=(torch.long, 0), object_id=(torch.long, 0), frame_index=(torch.long, 0))


@dataclass
class FindStage:
    img_ids: Any
    text_ids: Any
    input_boxes: Any
    input_boxes_mask: Any
    input_boxes_label: Any
    input_points: Any
    input_points_mask: Any
    object_ids: Optional[List[List]] = None


@dataclass
class BatchedFindTarget:
    num_boxes: Any
    boxes: Any
    boxes_padded: Any
    repeated_boxes: Any
    segments: Any
    semantic_segments: Any
    is_valid_segment: Any
    is_exhaustive: Any
    object_ids: Any
    object_ids_padded: Any


@dataclass
class BatchedInferenceMetadata:
    coco_image_id: Any
    original_image_id: Any
    original_category_id: Any
    original_size: Any
    object_id: Any
    frame_index: Any
    is_conditioning_only: List[Optional[bool]]


@dataclass
class BatchedDatapoint:
    img_batch: torch.Tensor
    find_text_batch: List[str]
    find_inputs: List[FindStage]
    find_targets: List[BatchedFindTarget]
    find_metadatas: List[BatchedInferenceMetadata]
    raw_images: Optional[List[Any]] = None


def _tensorise(record, spec: Dict[str, Tuple[torch.dtype, int]]):
    """Lists of tensors are stacked along the field's axis, lists of scalars become a tensor, None stays None."""
    for name, (dtype, axis) in spec.items():
        value = getattr(record, name)
        if value is None:
            continue
        if len(value) and isinstance(value[0], torch.Tensor):
            setattr(record, name, torch.stack(value, dim=axis).to(dtype))
        else:
            setattr(record, name, torch.as_tensor(value, dtype=dtype))
    return record


def _pad_to_longest(tensors: List[torch.Tensor], pad_val) -> List[torch.Tensor]:
    """Right-pad along axis 0 to the longest entry."""
    if not tensors:
        return tensors
    longest = max(t.shape[0] for t in tensors)
    out = []
    for t in tensors:
        pad = (0, 0) * (t.dim() - 1) + (0, longest - t.shape[0])
        out.append(torch.nn.functional.pad(t, pad, value=pad_val))
    return out


def _packed_to_padded(packed: torch.Tensor, counts: torch.Tensor, fill=0) -> torch.Tensor:
    """[sum n_i, ...] -> [B, max n_i, ...]."""
    ns = counts.tolist()
    out = packed.new_full((len(ns), max(ns), *packed.shape[1:]), fill)
    start = 0
    for i, n in enumerate(ns):
        out[i, :n] = packed[start:start + n]
        start += n
    return out


def collate_fn_api(batch: Sequence[Datapoint], dict_key, with_seg_masks: bool = False,
                   input_points_embedding_dim: int = 257, repeats: int = 0, load_image_in_fp16: bool = False) -> Dict:
    n_stages = max(q.query_processing_order for d in batch for q in d.find_queries) + 1
    stages = [FindStage([], [], [], [], [], [], [], object_ids=[]) for _ in range(n_stages)]
    targets = [BatchedFindTarget([], [], [], [], [], [], [], [], [], []) for _ in range(n_stages)]
    metas = [BatchedInferenceMetadata([], [], [], [], [], [], []) for _ in range(n_stages)]
    images: List[torch.Tensor] = []
    texts: List[str] = []
    raw_images = None
    first_image = 0
    for d in batch:
        images.extend(img.data for img in d.images)
        if d.raw_images is not None:
            raw_images = (raw_images or []) + list(d.raw_images)
        for q in d.find_queries:
            st, tg, md = stages[q.query_processing_order], targets[q.query_processing_order], metas[q.query_processing_order]
            st.img_ids.append(q.image_id + first_image)
            if q.query_text not in texts:
                texts.append(q.query_text)
            st.text_ids.append(texts.index(q.query_text))
            assert q.inference_metadata is not None, "inference_metadata must be provided when FindQueryLoaded is created."
            for f in fields(q.inference_metadata):
                getattr(md, f.name).append(getattr(q.inference_metadata, f.name))
            if q.input_bbox is not None:
                nb = q.input_bbox.numel() // 4
                assert q.input_bbox.numel() % 4 == 0 and q.input_bbox_label is not None and len(q.input_bbox_label) == nb
                st.input_boxes.append(q.input_bbox.view(nb, 4))
                st.input_boxes_label.append(q.input_bbox_label.view(nb))
                st.input_boxes_mask.append(torch.zeros(nb, dtype=torch.bool))
            else:
                st.input_boxes.append(torch.zeros(0, 4))
                st.input_boxes_label.append(torch.zeros(0, dtype=torch.bool))
                st.input_boxes_mask.append(torch.ones(0, dtype=torch.bool))
            if q.input_points is not None:
                st.input_points.append(q.input_points.squeeze(0))
                st.input_points_mask.append(torch.zeros(q.input_points.shape[1]))
            else:
                st.input_points.append(torch.empty(0, input_points_embedding_dim))
                st.input_points_mask.append(torch.empty(0))
            st.object_ids.append(q.object_ids_output)
            objects = [d.images[q.image_id].objects[i] for i in q.object_ids_output]
            boxes = [o.bbox for o in objects]
            tg.boxes.extend(boxes)
            tg.object_ids.extend(q.object_ids_output)
            for _ in range(repeats):
                tg.repeated_boxes.extend(boxes)
            tg.num_boxes.append(len(boxes))
            tg.is_exhaustive.append(q.is_exhaustive)
            if with_seg_masks:
                for o in objects:
                    if o.segment is not None:
                        tg.segments.append(o.segment)
                        tg.is_valid_segment.append(1)
                    else:
                        tg.segments.append(torch.zeros(d.images[q.image_id].data.shape[-2:], dtype=torch.bool))
                        tg.is_valid_segment.append(0)
            else:
                tg.segments = tg.is_valid_segment = None
            if q.semantic_target is not None:
                tg.semantic_segments.append(q.semantic_target)
        first_image += len(d.images)

    for i in range(n_stages):
        st, tg = stages[i], targets[i]
        st.input_points = _pad_to_longest(st.input_points, 0)
        st.input_points_mask = _pad_to_longest(st.input_points_mask, 1)
        st.input_boxes = _pad_to_longest(st.input_boxes, 0)
        st.input_boxes_label = _pad_to_longest(st.input_boxes_label, 0)
        st.input_boxes_mask = _pad_to_longest(st.input_boxes_mask, 1)
        # host-side note for the model (a plain attribute, not a field of the reference's record): every prompt reads
        # its own image, in order -> the per-prompt gather of the feature maps is the identity and can be skipped
        st.img_ids_are_arange = list(st.img_ids) == list(range(len(images)))
        _tensorise(st, _FIND_STAGE)
        tg.num_boxes_host = tuple(int(n) for n in tg.num_boxes)       # same kind of note: box counts, host side
        _tensorise(tg, _FIND_TARGET)
        _tensorise(metas[i], _METADATA)
        tg.boxes_padded = _packed_to_padded(tg.boxes.view(-1, 4), tg.num_boxes)
        tg.object_ids_padded = _packed_to_padded(tg.object_ids, tg.num_boxes, fill=-1)
    for img in images[1:]:
        assert img.shape == images[0].shape, "All images must have the same size"
    img_batch = torch.stack(images)
    if load_image_in_fp16:
        img_batch = img_batch.half()
    return {dict_key: BatchedDatapoint(img_batch=img_batch, find_text_batch=texts, find_inputs=stages,
                                       find_targets=targets, find_metadatas=metas, raw_images=raw_images)}


# ------------------------------------------------------------------------------------------------ synthetic data --
def synthetic_datapoint(idx: int, resolution: int = 1008, source: int = 1024, n_objects: int = 2,
                        text: str = "crack", seed: int = 1234) -> Datapoint:
    """One sample of SURVEY section 8(d).  Deterministic in (seed, idx)."""
    g = torch.Generator().manual_seed(seed + idx)
    raw = torch.rand(3, source, source, generator=g)
    if source != resolution:                    # PIL's BILINEAR resize is an antialiased triangle filter when shrinking
        raw = torch.nn.functional.interpolate(raw[None], size=(resolution, resolution), mode="bilinear",
                                              align_corners=False, antialias=True)[0]
    image = ((raw * 255.0).round().clamp(0, 255) / 255.0 - 0.5) / 0.5          # uint8 quantisation of the PIL path
    xy = torch.rand(n_objects, 2, generator=g) * 0.5
    wh = torch.rand(n_objects, 2, generator=g) * 0.3 + 0.1
    objects = []
    for i in range(n_objects):
        x1, y1 = xy[i].tolist()
        x2, y2 = min(x1 + wh[i, 0].item(), 1.0), min(y1 + wh[i, 1].item(), 1.0)
        box = torch.tensor([x1, y1, x2, y2], dtype=torch.float32)               # normalised xyxy, as the reference stores
        mask = torch.zeros(resolution, resolution, dtype=torch.bool)
        px = (box * resolution).round().long().tolist()
        mask[px[1]:max(px[3], px[1] + 1), px[0]:max(px[2], px[0] + 1)] = True
        objects.append(Object(bbox=box, area=(box[2] - box[0]) * (box[3] - box[1]), object_id=i, segment=mask))
    query = FindQueryLoaded(query_text=text, image_id=0, object_ids_output=list(range(n_objects)), is_exhaustive=True,
                            query_processing_order=0,
                            inference_metadata=InferenceMetadata(coco_image_id=idx, original_image_id=idx,
                                                                 original_category_id=0, original_size=(source, source),
                                                                 object_id=-1, frame_index=-1))
    return Datapoint(find_queries=[query], images=[Image(data=image, objects=objects, size=(resolution, resolution))])


class SyntheticSegmentDataset(torch.utils.data.Dataset):
    """``len`` samples of :func:`synthetic_datapoint` -- what ``training.data_dir: synthetic[:N]`` selects."""

    def __init__(self, length: int = 64, split: str = "train", resolution: int = 1008, source: int = 1024,
                 n_objects: int = 2, text: str = "crack", seed: int = 1234):
        self.length, self.resolution, self.source = length, resolution, source
        self.n_objects, self.text = n_objects, text
        self.seed = seed + (0 if split == "train" else 1_000_003)

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int) -> Datapoint:
        return synthetic_datapoint(idx, self.resolution, self.source, self.n_objects, self.text, self.seed)


# ----------------------------------------------------------------------------------------- rank-sharded loading --
def shard_indices(n: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0,
                  drop_last: bool = False) -> List[int]:
    """``torch.utils.data.DistributedSampler`` semantics (the reference's DDP trainer shards with it,
    ``sam3/train/data/torch_dataset.py:30-37``): a permutation seeded by ``seed + epoch`` shared by all ranks, padded
    by wrapping to a multiple of ``world`` (or truncated with ``drop_last``), rank ``r`` takes ``r, r + world, ...`` --
    so the shards are disjoint, equally long, and their union is the dataset."""
    if shuffle:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(seed + epoch)).tolist()
    else:
        order = list(range(n))
    if drop_last:
        order = order[:n // world * world]
    else:
        total = (n + world - 1) // world * world
        while len(order) < total:
            order += order[:total - len(order)]
    return order[rank::world]


class ShardedLoader:
    """Batches of a map-style dataset for one rank: ``shard_indices`` per epoch, ``collate_fn`` per batch -- the
    reference's ``DataLoader(dataset, batch_size, sampler=DistributedSampler(...), num_workers=0)`` (``num_workers=0`` is
    hard-coded there, ``train_sam3_lora_native.py:831``) with the sampler's ``set_epoch`` protocol; world size 1 = the
    reference's ``DataLoader(shuffle=...)``.

    ``num_workers > 0`` takes the data step off the training step's critical path without changing
    what is produced or its order: the samples of the next ``prefetch`` batches are built concurrently by a pool of
    threads (image decode / resize, mask rasterisation and the tensor arithmetic all release the GIL), collated in
    order by a producer thread and -- with ``device`` -- copied to the GPU from pinned memory on a side stream, so that
    ``next()`` normally returns a batch that is already resident.  ``stall_s`` accumulates the time ``next()`` had to
    wait (0 when the loader keeps up)."""

    def __init__(self, dataset, batch_size: int, collate_fn, shuffle: bool, rank: int = 0, world: int = 1, seed: int = 0,
                 num_workers: int = 0, prefetch: int = 2, device=None):
        self.dataset, self.batch_size, self.collate_fn = dataset, int(batch_size), collate_fn
        self.shuffle, self.rank, self.world, self.seed, self.epoch = shuffle, rank, world, seed, 0
        self.num_workers, self.prefetch = int(num_workers), max(1, int(prefetch))
        self.device = torch.device(device) if device is not None else None
        self.stall_s = 0.0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def indices(self) -> List[int]:
        return shard_indices(len(self.dataset), self.rank, self.world, self.epoch, self.shuffle, self.seed)

    def __len__(self) -> int:
        per_rank = (len(self.dataset) + self.world - 1) // self.world
        return (per_rank + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        idx = self.indices()
        groups = [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
        if self.num_workers <= 0:
            for g in groups:
                yield self.collate_fn([self.dataset[j] for j in g])
            return
        yield from self._iter_prefetched(groups)

    # ---- worker pool -------------------------------------------------------------------------------------------
    def _to_device(self, batch):
        """Pinned staging + asynchronous copy on a side stream; returns (batch on device, event to wait for)."""
        dev = self.device
        if dev is None or dev.type != "cuda":
            return batch, None
        if not hasattr(self, "_copy_stream"):
            self._copy_stream = torch.cuda.Stream(device=dev)

        def move(obj):
            if isinstance(obj, torch.Tensor):
                return (obj.pin_memory() if obj.numel() > 0 and not obj.is_pinned() else obj).to(dev, non_blocking=True)
            if isinstance(obj, list):
                return [move(x) for x in obj]
            if isinstance(obj, tuple):
                return tuple(move(x) for x in obj)
            if isinstance(obj, dict):
                return {k: move(v) for k, v in obj.items()}
            if hasattr(obj, "__dataclass_fields__"):
                for f in obj.__dataclass_fields__:
                    setattr(obj, f, move(getattr(obj, f)))
            return obj
        with torch.cuda.stream(self._copy_stream):
            out = move(batch)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return out, ev

    def _iter_prefetched(self, groups):
        import queue
        import threading
        import time
        from concurrent.futures import ThreadPoolExecutor
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        pool = ThreadPoolExecutor(max_workers=self.num_workers, thread_name_prefix="sam3-data")

        def producer():
            try:
                window = self.prefetch + 1          # batches whose samples are in flight in the pool
                futs = [[pool.submit(self.dataset.__getitem__, j) for j in g] for g in groups[:window]]
                for k in range(len(groups)):
                    if stop.is_set():
                        return
                    samples = [f.result() for f in futs[k]]
                    futs[k] = None
                    if k + window < len(groups):
                        futs.append([pool.submit(self.dataset.__getitem__, j) for j in groups[k + window]])
                    item = self._to_device(self.collate_fn(samples))
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as e:              # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=producer, name="sam3-data-producer", daemon=True)
        th.start()
        try:
            while True:
                t0 = time.perf_counter()
                item = q.get()
                self.stall_s += time.perf_counter() - t0
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    # The tensors were allocated on the copy stream's pool.  Tell the caching allocator that the CONSUMER's
                    # stream uses them: without this, dropping the batch hands the blocks straight back to the copy stream,
                    # and the producer's next H2D copy may overwrite memory that queued main-stream kernels (the loss
                    # backward's saved target masks) still read.
                    _record_stream(batch, cur)
                yield batch
        finally:
            stop.set()
            pool.shutdown(wait=False, cancel_futures=True)


def _record_stream(obj, stream) -> None:
    """``Tensor.record_stream(stream)`` on every CUDA tensor of a (nested) batch structure."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda and obj.numel() > 0:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for x in obj:
            _record_stream(x, stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif hasattr(obj, "__dataclass_fields__"):
        for f in obj.__dataclass_fields__:
            _record_stream(getattr(obj, f), stream)


# --------------------------------------------------------------------------------------------------- COCO data --
def rle_counts_from_string(s: Union[str, bytes]) -> List[int]:
    """COCO's compressed RLE string -> run lengths.  Published algorithm of ``pycocotools`` (``maskApi.c``
    ``rleFrString``; the reference imports pycocotools >= 2.0 un-pinned, ``train_sam3_lora_native.py:40``): each count
    is a little-endian base-32 varint offset by ASCII 48 with a continuation bit (0x20) and sign extension (0x10),
    and from the third count on it is stored as the difference to the count two places back."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts: List[int] = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(counts: Sequence[int], h: int, w: int) -> torch.Tensor:
    """Run lengths (column-major, starting with a run of zeros) -> bool mask [h, w]."""
    runs = torch.as_tensor(list(counts), dtype=torch.long)
    values = (torch.arange(len(runs)) % 2).bool()
    flat = torch.repeat_interleave(values, runs)
    if flat.numel() != h * w:
        raise ValueError(f"RLE covers {flat.numel()} pixels, expected {h * w}")
    return flat.view(w, h).t().contiguous()


def polygon_marks(xy: Sequence[float], h: int, w: int):
    """The run boundaries ``rleFrPoly`` derives from one polygon, as flat column-major positions (numpy int64, unsorted,
    duplicates kept).  Vectorised per edge; the scalar statement of the same rule is :func:`polygon_to_mask_loops`.
    A pixel of the mask is set when an odd number of marks lie at or before its column-major index -- which is what
    pycocotools' zero-length-run merging amounts to (equal marks cancel in pairs)."""
    import numpy as np
    k = len(xy) // 2
    scale = 5.0
    px = [int(scale * xy[2 * j] + 0.5) for j in range(k)]
    py = [int(scale * xy[2 * j + 1] + 0.5) for j in range(k)]
    px.append(px[0])
    py.append(py[0])
    us, vs = [], []
    for j in range(k):
        xs, xe, ys, ye = px[j], px[j + 1], py[j], py[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            slope = (ye - ys) / dx if dx else 0.0
            t = np.arange(dx, -1, -1, dtype=np.int64) if flip else np.arange(dx + 1, dtype=np.int64)
            us.append(t + xs)
            vs.append((ys + slope * t + 0.5).astype(np.int64))          # C cast: truncation, as int() in the scalar form
        else:
            slope = (xe - xs) / dy
            t = np.arange(dy, -1, -1, dtype=np.int64) if flip else np.arange(dy + 1, dtype=np.int64)
            vs.append(t + ys)
            us.append((xs + slope * t + 0.5).astype(np.int64))
    u, v = np.concatenate(us), np.concatenate(vs)
    u0, u1, v0, v1 = u[:-1], u[1:], v[:-1], v[1:]
    step = u1 != u0
    down = u1 < u0
    xd = (np.where(down, u1, u1 - 1).astype(np.float64) + 0.5) / scale - 0.5
    keep = step & (np.floor(xd) == xd) & (xd >= 0) & (xd <= w - 1)
    yd = (np.minimum(v0, v1).astype(np.float64) + 0.5) / scale - 0.5
    yd = np.ceil(np.clip(yd, 0.0, float(h)))
    return (xd[keep].astype(np.int64) * h + yd[keep].astype(np.int64))


def polygon_to_mask(xy: Sequence[float], h: int, w: int) -> torch.Tensor:
    """One polygon ``[x0, y0, x1, y1, ...]`` (pixels) -> bool mask [h, w] with ``pycocotools``' rasterisation rule
    (``maskApi.c`` ``rleFrPoly``, see :func:`polygon_to_mask_loops` for the rule itself): numpy form -- the marks of all
    edges at once, then the column-major parity fill.  Bit-identical to the scalar form (tests/test_sam3_data.py)."""
    import numpy as np
    marks = polygon_marks(xy, h, w)
    toggles = np.bincount(marks, minlength=h * w + 1)[:h * w] & 1
    flat = (np.cumsum(toggles, dtype=np.int64) & 1).astype(np.bool_)
    return torch.from_numpy(np.ascontiguousarray(flat.reshape(w, h).T))


def polygon_to_mask_loops(xy: Sequence[float], h: int, w: int) -> torch.Tensor:
    """One polygon ``[x0, y0, x1, y1, ...]`` (pixels) -> bool mask [h, w] with ``pycocotools``' rasterisation rule
    (``maskApi.c`` ``rleFrPoly``): vertices are scaled by 5 and rounded, every edge is walked on that fine grid,
    the crossings of the boundary with pixel-centre columns become the run boundaries of a column-major RLE (a
    pixel is inside when its centre lies to the right of an odd number of crossings above it).  Restated from the
    published algorithm; pycocotools is not installed in the build image, so this function is pinned by properties
    (exact rectangles, symmetry, area of known shapes), not by pycocotools' output: parity unpinned."""
    import math
    k = len(xy) // 2
    scale = 5.0
    px = [int(scale * xy[2 * j] + 0.5) for j in range(k)]
    py = [int(scale * xy[2 * j + 1] + 0.5) for j in range(k)]
    px.append(px[0])
    py.append(py[0])
    us: List[int] = []
    vs: List[int] = []
    for j in range(k):
        xs, xe, ys, ye = px[j], px[j + 1], py[j], py[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            slope = (ye - ys) / dx if dx else 0.0
            for d in range(dx + 1):
                t = dx - d if flip else d
                us.append(t + xs)
                vs.append(int(ys + slope * t + 0.5))
        else:
            slope = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                vs.append(t + ys)
                us.append(int(xs + slope * t + 0.5))
    marks: List[int] = []
    for j in range(1, len(us)):
        if us[j] == us[j - 1]:
            continue
        xd = float(us[j] if us[j] < us[j - 1] else us[j] - 1)
        xd = (xd + 0.5) / scale - 0.5
        if math.floor(xd) != xd or xd < 0 or xd > w - 1:
            continue
        yd = float(vs[j] if vs[j] < vs[j - 1] else vs[j - 1])
        yd = (yd + 0.5) / scale - 0.5
        yd = math.ceil(min(max(yd, 0.0), float(h)))
        marks.append(int(xd) * h + int(yd))
    marks.append(h * w)
    marks.sort()
    gaps, prev = [], 0
    for m in marks:
        gaps.append(m - prev)
        prev = m
    counts = [gaps[0]]
    j = 1
    while j < len(gaps):                # zero-length runs merge their neighbours
        if gaps[j] > 0:
            counts.append(gaps[j])
            j += 1
        else:
            j += 1
            if j < len(gaps):
                counts[-1] += gaps[j]
                j += 1
    return rle_decode(counts, h, w)


def segmentation_to_mask(segmentation, h: int, w: int) -> Optional[torch.Tensor]:
    """COCO ``segmentation`` field: RLE dict (compressed string or plain list of counts) or list of polygons (their
    union) -- ``train_sam3_lora_native.py:151-163``."""
    if isinstance(segmentation, dict):
        counts = segmentation["counts"]
        hh, ww = segmentation["size"]
        return rle_decode(rle_counts_from_string(counts) if isinstance(counts, (str, bytes)) else counts, hh, ww)
    if isinstance(segmentation, list):
        polys = segmentation if segmentation and isinstance(segmentation[0], (list, tuple)) else [segmentation]
        mask = torch.zeros(h, w, dtype=torch.bool)
        for poly in polys:
            if len(poly) >= 6:
                mask |= polygon_to_mask(poly, h, w)
        return mask
    return None


class COCOSegmentDataset(torch.utils.data.Dataset):
    """``<data_dir>/<split>/_annotations.coco.json`` + images -> :class:`Datapoint` (``train_sam3_lora_native.py:46-232``):
    image resized to 1008 x 1008 (PIL bilinear), scaled to [-1, 1]; per annotation a normalised xyxy box (the
    reference's format quirk) and a nearest-resized boolean mask; one query per image whose text is the (most common)
    category name, lower-cased, or "object" without annotations."""

    def __init__(self, data_dir, split: str = "train", resolution: int = 1008):
        import json
        from pathlib import Path
        self.data_dir, self.split = Path(data_dir), split
        self.split_dir = self.data_dir / split
        ann_file = self.split_dir / "_annotations.coco.json"
        if not ann_file.exists():
            raise FileNotFoundError(f"COCO annotation file not found: {ann_file}")
        with open(ann_file) as f:
            self.coco_data = json.load(f)
        self.images = {img["id"]: img for img in self.coco_data["images"]}
        self.image_ids = sorted(self.images)
        self.img_to_anns: Dict[int, List[dict]] = {}
        for ann in self.coco_data["annotations"]:
            self.img_to_anns.setdefault(ann["image_id"], []).append(ann)
        self.categories = {c["id"]: c["name"] for c in self.coco_data["categories"]}
        self.resolution = resolution
        print(f"Loaded COCO dataset: {split} split\n  Images: {len(self.image_ids)}\n"
              f"  Annotations: {len(self.coco_data['annotations'])}\n  Categories: {self.categories}")

    def __len__(self) -> int:
        return len(self.image_ids)

    def __getitem__(self, idx: int) -> Datapoint:
        import numpy as np
        from collections import Counter
        from PIL import Image as PILImage
        img_id = self.image_ids[idx]
        info = self.images[img_id]
        pil = PILImage.open(self.split_dir / info["file_name"]).convert("RGB")
        orig_w, orig_h = pil.size
        R = self.resolution
        pil = pil.resize((R, R), PILImage.BILINEAR)
        image = (torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5
        objects, names = [], []
        for i, ann in enumerate(self.img_to_anns.get(img_id, [])):
            bbox = ann.get("bbox")
            if bbox is None:
                continue
            names.append(self.categories.get(ann.get("category_id", 0), "object"))
            x, y, bw, bh = bbox
            # the reference's order of fp32 operations (:131-142): xyxy in pixels -> scaled to the model resolution -> divided
            # by it; a fused x / orig_w differs in the last bit (tests/golden/dataset_cases.npz pins the bits)
            box = torch.tensor([x, y, x + bw, y + bh], dtype=torch.float32)
            box[0] *= R / orig_w
            box[2] *= R / orig_w
            box[1] *= R / orig_h
            box[3] *= R / orig_h
            box /= R
            segment = None
            seg = ann.get("segmentation")
            if seg:
                try:
                    m = segmentation_to_mask(seg, orig_h, orig_w)
                    if m is None:
                        print(f"Warning: Unknown segmentation format: {type(seg)}")
                        continue
                    m = torch.nn.functional.interpolate(m[None, None].float(), size=(R, R), mode="nearest")
                    segment = m.squeeze() > 0.5
                except Exception as e:                  # a broken mask keeps the box, as in the reference
                    print(f"Warning: Error processing mask for image {img_id}, ann {i}: {e}")
            objects.append(Object(bbox=box, area=(box[2] - box[0]) * (box[3] - box[1]), object_id=i, segment=segment))
        if names:
            uniq = list(set(names))
            text = (uniq[0] if len(uniq) == 1 else Counter(names).most_common(1)[0][0]).lower()
        else:
            text = "object"
        query = FindQueryLoaded(query_text=text, image_id=0, object_ids_output=[o.object_id for o in objects],
                                is_exhaustive=True, query_processing_order=0,
                                inference_metadata=InferenceMetadata(coco_image_id=img_id, original_image_id=img_id,
                                                                     original_category_id=0,
                                                                     original_size=(orig_h, orig_w), object_id=-1,
                                                                     frame_index=-1))
        return Datapoint(find_queries=[query], images=[Image(data=image, objects=objects, size=(R, R))],
                         raw_images=[pil])
