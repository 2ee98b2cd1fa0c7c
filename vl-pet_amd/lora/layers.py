import torch.nn as nn


class LoRALayer:
    """Mixin holding the LoRA hyper-parameters (reference: lora/layers.py:12-29)."""

    def __init__(self, r: int, lora_alpha: int, lora_dropout: float, merge_weights: bool):
        self.r = r
        self.lora_alpha = lora_alpha
        self.lora_dropout_p = float(lora_dropout)
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else (lambda x: x)
        self.merged = False
        self.merge_weights = merge_weights
