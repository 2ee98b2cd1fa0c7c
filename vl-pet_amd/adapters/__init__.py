"""Drop-in for the reference's ``src/adapters`` operator package (hot-path subset): the same class
names, constructor arguments, ``forward(inputs, task, y=None)`` signature, config attribute names and
state-dict keys; the arithmetic runs in the fused HIP kernels.  Variants the kernels do not cover (other non-linearities, track_z,
low-rank adapters) run as plain torch ops (vl-pet_amd/eager.py: the eager fallback of SURVEY.md 8b); hyper-network and Compacter
(PHM) adapters are baselines outside the VL-PET path and are not provided."""
from .config import AdapterConfig
from .adapter_modeling import Adapter, LowRankAdapter, LowRankLinear
from .adapter_controller import AdapterController
from .adapter_utils import Activations

__all__ = ["AdapterConfig", "Adapter", "LowRankAdapter", "LowRankLinear", "AdapterController", "Activations"]
