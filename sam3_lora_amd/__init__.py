"""
sam3_lora_amd -- MI355X-native LoRA adapter path for SAM3 fine-tuning.

Holds only what the hot path needs:
  csrc/            hand-written gfx950 HIP kernels + the C-ABI of include/sam3_lora_amd.h
  _ffi, functional ctypes binding and the autograd node that calls it
  lora_layers      drop-in for the reference's top-level ``lora_layers`` module
  lora/            drop-in for the reference's ``sam3_lora.lora`` package
  ddp              flat-buffer all-reduce of the A/B gradients (RCCL over xGMI)
"""
__version__ = "0.1.0"

from .functional import LAYOUT_PACKAGE, LAYOUT_ROOT, lora_linear  # noqa: F401
