"""The CLI entry (train_sam3_lora_native.py) plus a dump of this rank's adapter tensors, for the 2-rank test."""
import os
import sys

import torch

from sam3_lora_amd import trainer as T

if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config"); ap.add_argument("--model-builder"); ap.add_argument("--data-builder")
    a = ap.parse_args()
    tr = T.SAM3TrainerNative(a.config, model_builder=T.resolve_builder(a.model_builder, "X", "model"),
                             data_builder=T.resolve_builder(a.data_builder, "X", "data"))
    tr.train()
    torch.save({n: p.detach().cpu() for n, p in tr.model.named_parameters() if p.requires_grad}, os.environ["TOY_DUMP"])
