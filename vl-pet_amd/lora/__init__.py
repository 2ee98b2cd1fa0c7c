"""Drop-in for the reference's ``src/lora`` package (the part its model files use):
``LoRALinearController`` + ``LoRALayer`` + ``LoraConfig``.  The copied loralib layers
(lora/layers.py:32-322 in the reference) are not referenced by any model file and are not provided."""
name = "lora"
from .config import LoraConfig
from .layers import LoRALayer
from .controller import LoRALinearController

__all__ = ["LoraConfig", "LoRALayer", "LoRALinearController"]
