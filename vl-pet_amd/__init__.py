"""vl-pet_amd: MI355X-native PET hot path for VL-PET-style fine-tuning.

Sub-modules are imported lazily; everything that computes goes through ``_lib`` and fails loudly when
``libvlpet_hip.so`` is missing.  (The numpy specification of the MFMA-fragment pack layouts that csrc/pack.hip
implements is test infrastructure: tests/packing_spec.py.)"""
__version__ = "0.1.0"
