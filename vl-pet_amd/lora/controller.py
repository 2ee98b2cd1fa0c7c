"""LoRA linear with per-task (or shared) A/B (reference: lora/controller.py:11-87)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import LoRALayer
from .. import functional as VF
from ..tail import _draw_seed

LINK_DELTA_GRAD = True      # (A/B switch, tools/ab_switches.py: False = autograd adds the two d/dx)


class LoRALinearController(nn.Linear, LoRALayer):
    """``F.linear(x, W, b) + (dropout(x) @ A[task].T @ B[task].T) * alpha/r``.

    The frozen base GEMM stays a library GEMM; the low-rank update, its scale and the add run in
    one fused HIP kernel (csrc/pet_fwd.hip, ACT_IDENTITY path).  ``weight``/``bias`` keep nn.Linear's
    names so the pretrained q_proj / v_proj tensors load into them."""

    def __init__(self, in_features: int, out_features: int, fan_in_fan_out: bool = False, config=None, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        self.tasks = config.tasks
        self.use_single_lora = config.use_single_lora
        LoRALayer.__init__(self, r=config.lora_dim, lora_alpha=config.lora_alpha,
                           lora_dropout=config.lora_dropout, merge_weights=True)
        self.fan_in_fan_out = fan_in_fan_out
        self.lora_As = nn.ParameterDict()
        self.lora_Bs = nn.ParameterDict()
        if self.r > 0:
            self.construct_lora_weights(self.tasks)
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        self.reset_parameters()
        if fan_in_fan_out:          # (lora/controller.py:46-47: the weight is stored transposed; such layers take the eager path below)
            self.weight.data = self.weight.data.T
        self._packs = {}

    def reset_parameters(self):
        nn.Linear.reset_parameters(self)
        if hasattr(self, "lora_As"):
            for task in self.tasks:
                nn.init.kaiming_uniform_(self.lora_As[task], a=math.sqrt(5))
                nn.init.zeros_(self.lora_Bs[task])

    def get_task(self, task):
        return task

    def construct_lora_weights(self, tasks):
        if self.use_single_lora:
            a = nn.Parameter(self.weight.new_zeros((self.r, self.in_features)))
            b = nn.Parameter(self.weight.new_zeros((self.out_features, self.r)))
            for task in tasks:
                self.lora_As[task] = a
                self.lora_Bs[task] = b
        else:
            for task in tasks:
                self.lora_As[task] = nn.Parameter(self.weight.new_zeros((self.r, self.in_features)))
                self.lora_Bs[task] = nn.Parameter(self.weight.new_zeros((self.out_features, self.r)))
        return self.lora_As, self.lora_Bs

    def forward(self, x, task):
        if self.in_features != self.out_features or self.fan_in_fan_out:
            # the fused delta kernel covers what the reference wraps (the square q_proj / v_proj of BART, my_transformers/
            # modeling_bart.py:767-768); any other layer shape takes the plain-torch composition ("else eager fallback")
            from .. import eager
            if self.r <= 0:
                return F.linear(x, (self.weight.t() if self.fan_in_fan_out else self.weight).to(x.dtype), None if self.bias is None else self.bias.to(x.dtype))
            return eager.lora_linear(x, self.weight, self.bias, self.lora_As[task], self.lora_Bs[task], self.scaling,
                                     float(self.lora_dropout_p), self.training, self.fan_in_fan_out)
        w, b = self.weight, self.bias
        if w.dtype != x.dtype:
            w = w.to(x.dtype)
        link = None
        if VF.linear_train_bias_ok(x, w, b):          # the bias trains (fp32 master) next to the frozen bf16 weight
            if self.r > 0 and LINK_DELTA_GRAD and x.requires_grad:
                link = VF.ResidualLink()              # K3's d/dx is taken over by this projection's dgrad GEMM (no autograd add)
            base = VF.linear_train_bias(x, w, b, link)
        else:
            if b is not None and b.dtype != x.dtype:
                b = b.to(x.dtype)
            base = F.linear(x, w, b)
        if self.r <= 0:
            return base
        A, B = self.lora_As[task], self.lora_Bs[task]
        cache = self._packs.setdefault(task, VF.PackCache())
        pk = cache.get([A], None, B, None, VF._io_dtype(x))
        # training-mode dropout of lora/controller.py:66: generated inside the kernels (one 64-bit seed per call, drawn
        # from torch's CPU generator so torch.manual_seed makes a run repeatable); the backward regenerates the mask
        p = float(self.lora_dropout_p) if self.training else 0.0
        seed = _draw_seed() if p > 0.0 else 0
        return VF.lora_delta(x, base, A, B, pk, self.scaling, None, p, seed, link=link)
