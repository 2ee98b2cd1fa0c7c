from dataclasses import dataclass
from typing import List, Optional


@dataclass
class LoraConfig:
    """reference: lora/config.py:4-8 plus the fields trainer_base.py:203-208 assigns."""
    lora_dim: int = 4
    lora_alpha: int = 32
    lora_dropout: float = 0.1
    tasks: Optional[List[str]] = None
    use_single_lora: bool = False
