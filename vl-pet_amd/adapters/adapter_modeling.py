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
        self.activation = Activations(config.non_linearity.lower())
        # the fused kernels implement gelu_new and never materialise the [M, r] bottleneck: any other non-linearity, or track_z
        # (multitask.py:246-249 reads adapter.z), runs the plain-torch composition of vl-pet_amd/eager.py ("else eager
        # fallback"); no launch script of the reference asks for either
        self.eager = config.non_linearity.lower() != "gelu_new" or bool(config.track_z)
        self.down_sampler = nn.Linear(self.input_dim, self.down_sample_size)
        self.up_sampler = nn.Linear(self.down_sample_size, self.input_dim)
        self.track_z = config.track_z
        self._pack = VF.PackCache()

    def packed(self, io_dtype):
        return self._pack.get([self.down_sampler.weight], [self.down_sampler.bias], self.up_sampler.weight,
                              self.up_sampler.bias, io_dtype)

    def fused(self, x, residual, scale=1.0, link=None):
        """residual + scale * adapter(x) in one kernel (K2).  ``link``: functional.parallel_adapter."""
        if self.eager:
            out = self.forward(x)
            return residual + (out if scale == 1.0 else scale * out)
        pk = self.packed(VF._io_dtype(x))
        return VF.parallel_adapter(x, residual, self.down_sampler.weight, self.down_sampler.bias,
                                   self.up_sampler.weight, self.up_sampler.bias, pk, scale, link=link)

    def _keep_z(self, z):
        if self.track_z:
            self.z = z

    def forward(self, x):
        if self.eager:
            from .. import eager
            return eager.adapter(x, self.down_sampler.weight, self.down_sampler.bias, self.up_sampler.weight, self.up_sampler.bias,
                                 self.activation, self._keep_z)
        # bare adapter output (no residual): K1 kernel with gate off and x2_scale = 0
        pk = self.packed(VF._io_dtype(x))
        return VF.adapter_gate(None, x, [self.down_sampler.weight], [self.down_sampler.bias], self.up_sampler.weight,
                               self.up_sampler.bias, None, pk, None, VF.GATE_NONE, 1.0, 0.0, 1.0)


class LowRankLinear(nn.Module):
    """``x @ (W_left @ W_right) + b`` with rank-``rank`` factors (reference: adapters/low_rank_layer.py:8-46; parameter names kept)."""

    def __init__(self, input_dim: int, output_dim: int, rank: int = 1, bias: bool = True, w_init: str = "glorot-uniform"):
        super().__init__()
        self.input_dim, self.output_dim, self.rank, self.bias, self.w_init = input_dim, output_dim, rank, bias, w_init
        self.W_left = nn.Parameter(torch.empty(input_dim, rank))
        self.W_right = nn.Parameter(torch.empty(rank, output_dim))
        if bias:
            self.b = nn.Parameter(torch.zeros(output_dim))
        self.reset_parameters()

    def reset_parameters(self):
        if self.bias:
            nn.init.zeros_(self.b)
        if self.w_init == "glorot-uniform":
            nn.init.xavier_uniform_(self.W_left); nn.init.xavier_uniform_(self.W_right)
        elif self.w_init == "glorot-normal":
            nn.init.xavier_normal_(self.W_left); nn.init.xavier_normal_(self.W_right)
        else:
            raise ValueError(f"unknown low-rank init {self.w_init!r}")

    def forward(self, x):
        out = x @ (self.W_left @ self.W_right).to(x.dtype)
        return out + self.b.to(x.dtype) if self.bias else out


class LowRankAdapter(nn.Module):
    """Adapter whose two projections are low-rank products (reference: adapters/adapter_modeling.py:9-33).  One of the reference's
    baselines (outside the VL-PET hot path), so it runs as plain torch ops -- the eager fallback of SURVEY 8(b)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.input_dim = config.input_dim
        self.down_sample_size = self.input_dim // config.reduction_factor
        self.activation = Activations(config.non_linearity.lower())
        self.down_sampler = LowRankLinear(self.input_dim, self.down_sample_size, w_init=config.low_rank_w_init, rank=config.low_rank_rank)
        self.up_sampler = LowRankLinear(self.down_sample_size, self.input_dim, w_init=config.low_rank_w_init, rank=config.low_rank_rank)
        self.track_z = config.track_z
        self.eager = True

    def forward(self, x):
        z = self.activation(self.down_sampler(x))
        if self.track_z:
            self.z = z
        return self.up_sampler(z)

    def fused(self, x, residual, scale=1.0, link=None):
        out = self.forward(x)
        return residual + (out if scale == 1.0 else scale * out)
