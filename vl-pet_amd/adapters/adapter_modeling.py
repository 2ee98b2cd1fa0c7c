"""Bottleneck adapter (reference: adapters/adapter_modeling.py:36-61)."""
import torch
import torch.nn as nn

from .adapter_utils import Activations
from .. import functional as VF


class Adapter(nn.Module):
    """``up_sampler(act(down_sampler(x)))``.  Parameters are ordinary ``nn.Linear`` modules so the
    reference's checkpoints, init and name-substring freeze rules apply unchanged; the forward runs
    the fused HIP kernel (csrc/pet_fwd.hip)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.input_dim = config.d_model
        if config.use_adapter_down_dim:
            self.down_sample_size = config.adapter_down_dim
        else:
            self.down_sample_size = self.input_dim // config.reduction_factor
        if config.non_linearity.lower() != "gelu_new":
            raise NotImplementedError("the fused adapter kernel implements gelu_new only "
                                      f"(got {config.non_linearity!r})")
        self.activation = Activations(config.non_linearity.lower())
        self.down_sampler = nn.Linear(self.input_dim, self.down_sample_size)
        self.up_sampler = nn.Linear(self.down_sample_size, self.input_dim)
        self.track_z = config.track_z
        self._pack = VF.PackCache()

    def packed(self, io_dtype):
        return self._pack.get([self.down_sampler.weight], [self.down_sampler.bias], self.up_sampler.weight,
                              self.up_sampler.bias, io_dtype)

    def fused(self, x, residual, scale=1.0, link=None):
        """residual + scale * adapter(x) in one kernel (K2).  ``link``: functional.parallel_adapter."""
        if self.track_z:
            raise NotImplementedError("track_z needs the [M,r] bottleneck materialised; the fused path never writes it")
        pk = self.packed(VF._io_dtype(x))
        return VF.parallel_adapter(x, residual, self.down_sampler.weight, self.down_sampler.bias,
                                   self.up_sampler.weight, self.up_sampler.bias, pk, scale, link=link)

    def forward(self, x):
        # bare adapter output (no residual): K1 kernel with gate off and x2_scale = 0
        pk = self.packed(VF._io_dtype(x))
        return VF.adapter_gate(None, x, [self.down_sampler.weight], [self.down_sampler.bias], self.up_sampler.weight,
                               self.up_sampler.bias, None, pk, None, VF.GATE_NONE, 1.0, 0.0, 1.0)
