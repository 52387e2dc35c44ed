"""Top-level name the reference CLI imports (``from lora_layers import ...``); the
implementation lives in ``sam3_lora_amd.lora_layers`` (HIP path)."""
from sam3_lora_amd.lora_layers import *  # noqa: F401,F403
from sam3_lora_amd.lora_layers import __all__  # noqa: F401
