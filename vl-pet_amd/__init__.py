"""vl-pet_amd: MI355X-native PET hot path for VL-PET-style fine-tuning.

Sub-modules are imported lazily so that the pure-host pieces (``packing``) work without torch
or the HIP library; everything that computes goes through ``_lib`` and fails loudly when
``libvlpet_hip.so`` is missing."""
__version__ = "0.1.0"
