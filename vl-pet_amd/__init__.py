"""vl-pet_amd: MI355X-native PET hot path for VL-PET-style fine-tuning.

Sub-modules are imported lazily; everything that computes goes through ``_lib`` and fails loudly when
``libvlpet_hip.so`` is missing.  (The numpy specification of the MFMA-fragment pack layouts that csrc/pack.hip
implements is test infrastructure: tests/packing_spec.py.)"""
import os as _os

# Kernel arguments in device memory: every kernel of the path starts with scalar loads from the kernel-argument segment, and
# with the runtime's default (host memory) placement each dependent level of them costs ~1.5-2 us per launch
# (profiles/r02_wgrad_stream_probes.md: 20,261 -> 20,950 samples/s on the same box).  Read by the HIP runtime when it
# initialises, so it has to be set before the first HIP call of the process; an explicit setting of the user wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
