"""MI355X-native drop-in for the reference package ``sam3_lora.lora``."""
from .lora_layer import LinearWithLoRA, LoRALayer
from .lora_utils import (LoRAConfig, get_lora_parameters, get_lora_state_dict, inject_lora_into_model,
                         load_lora_state_dict, merge_lora_weights, print_trainable_parameters)

__all__ = ["LoRALayer", "LinearWithLoRA", "LoRAConfig", "inject_lora_into_model", "get_lora_parameters",
           "get_lora_state_dict", "load_lora_state_dict", "merge_lora_weights", "print_trainable_parameters"]
