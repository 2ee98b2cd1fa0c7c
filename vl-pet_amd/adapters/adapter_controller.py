"""Task-dispatching adapter controller (reference: adapters/adapter_controller.py:11-162)."""
import torch
import torch.nn as nn

from .adapter_modeling import Adapter, LowRankAdapter


class AdapterController(nn.Module):
    """Holds one adapter per task (or one shared object registered under every task name when
    ``use_single_adapter``) and applies it with a sequential (``+ inputs``) or parallel (``+ y``)
    residual.  Same constructor / forward contract as the reference class."""

    def __init__(self, config):
        super().__init__()
        if getattr(config, "hypercomplex_adapters", False):
            # Compacter's PHM layers (adapters/hypercomplex/): a baseline of the reference, out of scope per SURVEY.md sections 1 / 2
            raise NotImplementedError("hypercomplex (PHM / Compacter) adapters are a baseline outside the VL-PET hot path (SURVEY.md section 2)")
        self.config = config
        self.low_rank_adapters = bool(getattr(config, "low_rank_adapters", False))     # eager fallback: adapter_modeling.LowRankAdapter
        self.hypercomplex_adapters = False
        self.tasks = config.tasks
        self.use_single_adapter = config.use_single_adapter
        self.share_up_sampler = getattr(config, "share_up_sampler", False)
        self.share_down_sampler = getattr(config, "share_down_sampler", False)
        self.adapters = self.construct_adapters(self.tasks)
        self.add_layer_norm_before_adapter = config.add_layer_norm_before_adapter
        self.add_layer_norm_after_adapter = config.add_layer_norm_after_adapter
        if self.add_layer_norm_before_adapter:
            self.pre_layer_norm = nn.LayerNorm(config.input_dim)
        if self.add_layer_norm_after_adapter:
            self.post_layer_norm = nn.LayerNorm(config.input_dim)

    def get_task(self, task):
        return task

    def construct_adapters(self, tasks):
        adapters = nn.ModuleDict()
        Adapter_ = LowRankAdapter if self.low_rank_adapters else Adapter
        if self.use_single_adapter:
            shared = Adapter_(self.config)
            for task in tasks:
                adapters[task] = shared
        else:
            for task in tasks:
                adapters[task] = Adapter_(self.config)
            if self.share_up_sampler:
                for task in tasks:
                    adapters[task].up_sampler = adapters[tasks[0]].up_sampler
            if self.share_down_sampler:
                for task in tasks:
                    adapters[task].down_sampler = adapters[tasks[0]].down_sampler
        return adapters

    @staticmethod
    def convert_to_list(tasks):
        return tasks if isinstance(tasks, list) else [tasks]

    def get_adapter(self, task):
        return self.adapters[task]

    def enable_adapters(self, tasks):
        for task in self.convert_to_list(tasks):
            for p in self.get_adapter(task).parameters():
                p.requires_grad = True

    def disable_adapters(self, tasks):
        for task in self.convert_to_list(tasks):
            for p in self.get_adapter(task).parameters():
                p.requires_grad = False

    def forward(self, inputs, task, y=None, link=None):
        # (`link`, not in the reference's signature: functional.parallel_adapter -- the frozen projection that made `y`
        #  from `inputs` takes over the adapter's input gradient; only when the adapter reads `inputs` itself)
        adapter = self.get_adapter(self.get_task(task))
        z = self.pre_layer_norm(inputs) if self.add_layer_norm_before_adapter else inputs
        scale = float(self.config.scaling_factor) if self.config.use_scaling_factor else 1.0
        residual = y if self.config.use_parallel_adapter else inputs
        if self.config.use_parallel_adapter and y is None:
            raise ValueError("use_parallel_adapter needs the parallel branch output `y`")
        if self.add_layer_norm_after_adapter:
            outputs = adapter(z)
            if scale != 1.0:
                outputs = scale * outputs
            return self.post_layer_norm(outputs) + residual
        if link is None or z is not inputs:
            return adapter.fused(z, residual, scale)
        return adapter.fused(z, residual, scale, link=link)
