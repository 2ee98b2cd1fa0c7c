"""Drop-in for the reference's ``src/adapters`` operator package (hot-path subset): the same class
names, constructor arguments, ``forward(inputs, task, y=None)`` signature, config attribute names and
state-dict keys; the arithmetic runs in the fused HIP kernels.  Hyper-network, Compacter (PHM) and
rank-1 low-rank adapters are baselines outside the VL-PET path and are not provided."""
from .config import AdapterConfig
from .adapter_modeling import Adapter
from .adapter_controller import AdapterController
from .adapter_utils import Activations

__all__ = ["AdapterConfig", "Adapter", "AdapterController", "Activations"]
