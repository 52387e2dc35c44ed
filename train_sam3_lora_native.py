#!/usr/bin/env python3
"""
Drop-in for the reference's training entry point (``train_sam3_lora_native.py`` there, ``main`` :1049-1060):

    python train_sam3_lora_native.py --config configs/full_lora_config.yaml \
        --model-builder my_pkg.builders:sam3_image_model --data-builder my_pkg.builders:coco_batches
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 train_sam3_lora_native.py --config ... (one rank per GPU)

Same ``--config`` flag, default and YAML keys; the LoRA adapters run on the gfx950 kernels.  See
sam3_lora_amd/trainer.py for what the two builders must return.
"""
from sam3_lora_amd.trainer import main

if __name__ == "__main__":
    main()
