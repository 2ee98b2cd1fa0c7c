import torch
import torch.nn as nn

from ..activations import get_activation


class Activations(nn.Module):
    """Named activation (reference: adapters/adapter_utils.py:7-13)."""

    def __init__(self, activation_type: str):
        super().__init__()
        self.name = activation_type
        self.f = get_activation(activation_type)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.f(x)
