from sam3_lora_amd.lora import *  # noqa: F401,F403
from sam3_lora_amd.lora import __all__  # noqa: F401
