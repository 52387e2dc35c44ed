#!/usr/bin/env python3
"""Validation of trained adapters over a COCO split (the reference's ``validate_sam3_lora.py`` CLI and flags):

    python validate_sam3_lora.py --config configs/full_lora_config.yaml --weights outputs/sam3_lora_full/best_lora_weights.pt \
        --val_data_dir /data/valid [--merge] [--num-samples N] [--prob-threshold 0.3] [--nms-iou 0.7] [--use-base-model]"""
import argparse

from sam3_lora_amd.inference import validate


def main(argv=None):
    ap = argparse.ArgumentParser(description="Validate a SAM3 LoRA model: mask AP with mask NMS")
    ap.add_argument("--config", default=None)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--val_data_dir", required=True, help="directory holding _annotations.coco.json and the images")
    ap.add_argument("--use-base-model", action="store_true")
    ap.add_argument("--num-samples", type=int, default=None)
    ap.add_argument("--prob-threshold", type=float, default=0.3)
    ap.add_argument("--nms-iou", type=float, default=0.7)
    ap.add_argument("--merge", action="store_true", help="merge overlapping segments (crack detection)")
    ap.add_argument("--merge-iou", type=float, default=0.15)
    a = ap.parse_args(argv)
    if not a.use_base_model and (a.config is None or a.weights is None):
        ap.error("--config and --weights are required when not using --use-base-model")
    validate(a.config, a.weights, a.val_data_dir, a.num_samples, a.prob_threshold, a.nms_iou, a.merge, a.merge_iou,
             a.use_base_model)


if __name__ == "__main__":
    main()
