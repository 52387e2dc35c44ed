#!/usr/bin/env python3
"""SAM3 + LoRA inference with adapters trained by train_sam3_lora_native.py (the reference's ``infer_sam.py`` CLI):

    python infer_sam.py --config configs/full_lora_config.yaml --image path/to/image.jpg --prompt crack defect --output out.png

Options beyond the reference: ``--merge`` folds the adapters into the frozen Linears first (adapter-free model)."""
import argparse

from sam3_lora_amd.inference import SAM3LoRAInference


def main(argv=None):
    ap = argparse.ArgumentParser(description="SAM3 + LoRA inference (MI355X adapter path)")
    ap.add_argument("--config", required=True, help="training config YAML")
    ap.add_argument("--weights", default=None, help="LoRA weights (default: <output_dir>/best_lora_weights.pt)")
    ap.add_argument("--image", required=True)
    ap.add_argument("--prompt", nargs="+", default=["object"], help="text prompt(s)")
    ap.add_argument("--output", default="output.png")
    ap.add_argument("--threshold", type=float, default=0.5)
    ap.add_argument("--resolution", type=int, default=1008)
    ap.add_argument("--no-boxes", action="store_true")
    ap.add_argument("--no-masks", action="store_true")
    ap.add_argument("--merge", action="store_true", help="merge the adapters into the base weights before inference")
    a = ap.parse_args(argv)
    inf = SAM3LoRAInference(a.config, a.weights, resolution=a.resolution, detection_threshold=a.threshold, merge=a.merge)
    res = inf.predict(a.image, a.prompt)
    n = inf.visualize(res, a.output, show_boxes=not a.no_boxes, show_masks=not a.no_masks)
    print(f"{n} detections drawn -> {a.output}")


if __name__ == "__main__":
    main()
