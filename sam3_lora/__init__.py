"""Import-compatible alias of the reference package name: ``sam3_lora.lora`` resolves to the
MI355X implementation in ``sam3_lora_amd.lora``."""
__version__ = "0.1.0"

from .lora import LinearWithLoRA, LoRAConfig, LoRALayer, inject_lora_into_model  # noqa: F401

__all__ = ["LoRALayer", "LinearWithLoRA", "LoRAConfig", "inject_lora_into_model"]
