"""Activation lookup.  ``gelu_new`` is the tanh form of HF's NewGELUActivation, the only
non-linearity the PET hot path uses (adapters/config.py:10, my_transformers/modeling_bart.py:1000,1044);
the fused kernels implement it in-register -- this torch version exists for the host model's frozen
parts and for module attributes that mirror the reference."""
import math

import torch
import torch.nn.functional as F


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


_ACT = {"gelu_new": gelu_new, "gelu": F.gelu, "relu": F.relu, "tanh": torch.tanh, "identity": lambda x: x}


def get_activation(name: str):
    try:
        return _ACT[name.lower()]
    except KeyError:
        raise KeyError(f"unknown activation {name!r}; have {sorted(_ACT)}")
