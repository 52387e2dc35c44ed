"""
Golden vectors for the consumers of the adapter files: the post-processing of the reference's validation script,

    apply_sam3_nms            validate_sam3_lora.py:303-352  (score filter -> sam3.perflib.nms.nms_masks -> top-k)
    merge_overlapping_masks   validate_sam3_lora.py:232-300  (greedy union of overlapping masks, best score first)

produced by the REFERENCE's own functions: the two function definitions are taken out of /root/reference/validate_sam3_lora.py
with `ast` at run time and executed unmodified (the script's module-level imports pull in pycocotools / torchvision / the
SAM3 evaluators, which this image lacks), `nms_masks` is imported from /root/reference/sam3/perflib/nms.py (CPU path,
generic_nms_cpu).  Only inputs' seeds and outputs are stored -> consumer_cases.npz.  Build container only.

    python tests/golden/make_consumer_golden.py
"""
import ast
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
import consumer_case_defs as C


def reference_functions():
    sam3_manifest._install_stubs()
    from sam3.perflib.nms import nms_masks
    src = open(os.path.join(REF, "validate_sam3_lora.py")).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "nms_masks": nms_masks}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("apply_sam3_nms", "merge_overlapping_masks"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "validate_sam3_lora.py", "exec"), ns)
    return ns["apply_sam3_nms"], ns["merge_overlapping_masks"]


def main():
    nms, merge = reference_functions()
    res = {}
    for name, n, side, clusters, seed, pt, it, k, mt in C.CASES:
        logits, masks, boxes = C.make_case(n, side, clusters, seed)
        fm, fs, fb = nms(logits, masks, boxes, prob_threshold=pt, nms_iou_threshold=it, max_detections=k)
        res[f"{name}/nms_masks"], res[f"{name}/nms_scores"], res[f"{name}/nms_boxes"] = fm.numpy(), fs.numpy(), fb.numpy()
        # the validation script merges the thresholded survivors (validate_sam3_lora.py:417-425)
        if len(fm) > 0:
            mm, ms, mb = merge(fm > 0.5, fs, fb, iou_threshold=mt)
        else:
            mm, ms, mb = fm > 0.5, fs, fb
        res[f"{name}/merged_masks"] = np.packbits(mm.numpy().astype(bool), axis=-1)
        res[f"{name}/merged_scores"], res[f"{name}/merged_boxes"] = ms.numpy(), mb.numpy()
        res[f"{name}/merged_count"] = np.int64(len(mm))
        print(f"{name}: {n} detections -> {len(fs)} after NMS -> {len(mm)} after merging")
    np.savez_compressed(os.path.join(HERE, "consumer_cases.npz"), **res)
    print("bytes:", os.path.getsize(os.path.join(HERE, "consumer_cases.npz")))


if __name__ == "__main__":
    main()
