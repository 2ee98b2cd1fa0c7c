"""Adapter configuration: the attribute names ``AdapterController`` / ``Adapter`` read
(reference: adapters/config.py:4-55 and the assignments in trainer_base.py:141-178)."""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class AdapterConfig:
    # bottleneck
    non_linearity: str = "gelu_new"
    reduction_factor: int = 16
    use_adapter_down_dim: bool = False
    adapter_down_dim: int = 96
    input_dim: int = 768
    d_model: int = 768
    # controller
    tasks: Optional[List[str]] = None
    use_single_adapter: bool = False
    share_up_sampler: bool = False
    share_down_sampler: bool = False
    add_layer_norm_before_adapter: bool = False
    add_layer_norm_after_adapter: bool = False
    use_parallel_adapter: bool = False
    use_scaling_factor: bool = False
    scaling_factor: float = 1.0
    track_z: bool = False
    # variants outside the hot path: low-rank adapters run as plain torch ops (eager fallback), hypercomplex (PHM) ones are refused
    low_rank_adapters: bool = False
    low_rank_w_init: str = "glorot-uniform"
    low_rank_rank: int = 1
    hypercomplex_adapters: bool = False
    shared_phm_rule: bool = False
    shared_phm_rule_over_tasks: bool = False
