"""ctypes binding of libvlpet_hip.so (the C ABI in include/vlpet_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, a
RuntimeError is raised."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLPET_LIB") or os.path.join(_HERE, "lib", "libvlpet_hip.so")   # override: same-box A/B of two builds

VLPET_F32 = 0
VLPET_BF16 = 1
GATE_NONE, GATE_MUL, GATE_ADD = 0, 1, 2

_lib = None

# name -> (restype, argtypes); one row per declaration in include/vlpet_hip.h
SIGNATURES = {
    "vlpet_version": (c_int, []),
    "vlpet_error_string": (c_char_p, [c_int]),
    "vlpet_rank_tiles": (c_int, [c_int]),
    "vlpet_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vlpet_pack_pair": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p]),
    "vlpet_pack_pairs": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p]),
    "vlpet_adapter_gate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                       c_int, c_float, c_float, c_float, c_int, c_void_p]),
    "vlpet_saved_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vlpet_adapter_gate_fwd_save": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                            c_int, c_int, c_float, c_float, c_float, c_int, c_void_p]),
    "vlpet_adapter_gate_bwd_saved": (c_int, [c_int] + [c_void_p] * 8 + [c_void_p] * 8 + [c_int, c_int, c_void_p, c_size_t,
                                                                                        c_int64, c_int, c_int, c_int,
                                                                                        c_float, c_float, c_float, c_int,
                                                                                        c_void_p]),
    "vlpet_adapter_gate_bwd_saved_acc": (c_int, [c_int] + [c_void_p] * 9 + [c_void_p] * 8 + [c_int, c_int, c_void_p, c_size_t,
                                                                                            c_int64, c_int, c_int, c_int,
                                                                                            c_float, c_float, c_float, c_int,
                                                                                            c_void_p]),
    "vlpet_adapter_gate_bwd_saved_y": (c_int, [c_int] + [c_void_p] * 10 + [c_void_p] * 8 + [c_int, c_int, c_void_p, c_size_t,
                                                                                           c_int64, c_int, c_int, c_int,
                                                                                           c_float, c_float, c_float, c_int,
                                                                                           c_void_p]),
    "vlpet_finalize_defer": (c_int, [c_int]),
    "vlpet_finalize_pending": (c_int, []),
    "vlpet_finalize_discard": (c_int, []),
    "vlpet_finalize_flush": (c_int, [c_void_p]),
    "vlpet_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "vlpet_adapter_gate_bwd_form": (c_int, [c_int64, c_int, c_int, c_int]),
    "vlpet_adapter_gate_bwd_finalize_launch": (c_int, [c_int64, c_int, c_int, c_int]),
    "vlpet_set_in_launch_reduce": (c_int, [c_int]),
    "vlpet_debug_build": (c_int, []),
    "vlpet_test_hold_cus": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p]),
    "vlpet_set_seed_counter": (c_int, [c_void_p]),
    "vlpet_adapter_gate_bwd": (c_int, [c_void_p] * 7 + [c_void_p] * 8 + [c_int, c_int, c_void_p, c_size_t, c_int64,
                                                                         c_int, c_int, c_int, c_float, c_float,
                                                                         c_float, c_int, c_void_p]),
    "vlpet_adapter_gate_bwd_phase": (c_int, [c_int] + [c_void_p] * 7 + [c_void_p] * 8 + [c_int, c_int, c_void_p, c_size_t,
                                                                                        c_int64, c_int, c_int, c_int,
                                                                                        c_float, c_float, c_float, c_int,
                                                                                        c_void_p]),
    "vlpet_parallel_adapter_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float,
                                           c_int, c_void_p]),
    "vlpet_parallel_adapter_bwd": (c_int, [c_void_p] * 4 + [c_void_p] * 4 + [c_int, c_void_p, c_size_t, c_int64,
                                                                             c_int, c_int, c_float, c_int, c_void_p]),
    "vlpet_parallel_adapter_fwd_save": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                                c_float, c_int, c_void_p]),
    "vlpet_parallel_adapter_bwd_saved": (c_int, [c_void_p] * 5 + [c_void_p] * 4 + [c_int, c_void_p, c_size_t, c_int64,
                                                                                   c_int, c_int, c_float, c_int, c_void_p]),
    "vlpet_lora_delta_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_uint64, c_void_p, c_void_p,
                                     c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "vlpet_lora_saved_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "vlpet_lora_delta_fwd_save": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_uint64, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "vlpet_lora_r8_applies": (c_int, [c_int64, c_int, c_int, c_int]),
    "vlpet_lora_delta_fwd_r8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_uint64, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "vlpet_lora_delta_bwd_saved": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_uint64, c_void_p,
                                           c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int64, c_int, c_int, c_float,
                                           c_int, c_void_p]),
    "vlpet_visproj_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vlpet_visproj_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlpet_visproj_fwd": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vlpet_visproj_gemm_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vlpet_visproj_gemm_exchange_bytes": (c_size_t, [c_int]),
    "vlpet_visproj_fwd_gemm": (c_int, [c_void_p] * 11 + [c_size_t, c_int64, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vlpet_visproj_fwd_gemm_cfg": (c_int, [c_void_p] * 11 + [c_size_t, c_int64, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_void_p]),
    "vlpet_sum_n": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "vlpet_vispos_applies": (c_int, [c_int, c_int]),
    "vlpet_vispos_fwd": (c_int, [c_void_p] * 5 + [c_void_p, c_int, c_int, c_void_p, c_int64] + [c_void_p, c_int, c_int64, c_void_p, c_int64]
                         + [c_void_p, c_int64, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vlpet_vispos_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vlpet_vispos_bwd": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_int64] + [c_void_p] * 5
                         + [c_void_p, c_size_t, c_int64, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vlpet_visproj_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vlpet_visproj_wgrad_workspace_bytes_io": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "vlpet_visproj_wgrad": (c_int, [c_void_p] * 5 + [c_size_t, c_int64, c_int, c_int, c_int, c_void_p]),
    "vlpet_lowrank_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vlpet_lowrank_pack": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]),
    "vlpet_lowrank_gate_fwd": (c_int, [c_void_p] * 5 + [c_int64] + [c_int] * 5 + [c_void_p]),
    "vlpet_lowrank_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "vlpet_lowrank_gate_bwd": (c_int, [c_void_p] * 13 + [c_int, c_int, c_void_p, c_size_t, c_int64] + [c_int] * 5 + [c_void_p]),
    "vlpet_norm_residual_fwd": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_float, c_int, c_void_p]),
    "vlpet_rowgate_partials": (c_int, [c_int64]),
    "vlpet_row_dot": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "vlpet_row_affine": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p]),
    "vlpet_rowgate_bwd": (c_int, [c_void_p] * 10 + [c_int64, c_int, c_int, c_void_p]),
    "vlpet_vecgate_fwd": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p]),
    "vlpet_vecgate_bwd": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "vlpet_optim_blocks": (c_int, [c_int64]),
    "vlpet_grad_sumsq": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "vlpet_adamw_step": (c_int, [c_void_p] * 5 + [c_int64, c_void_p, c_int] + [c_float] * 7 + [c_int, c_int, c_int,
                                                                                           c_void_p, c_void_p]),
    "vlpet_adamw_step_sliced": (c_int, [c_void_p] * 5 + [c_int64, c_void_p, c_int] + [c_float] * 7 + [c_void_p, c_void_p,
                                                                                                  c_int, c_int, c_void_p,
                                                                                                  c_void_p]),
    "vlpet_downsample_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vlpet_sublayer_tail_partials": (c_int, [c_int64]),
    "vlpet_sublayer_tail_fwd": (c_int, [c_void_p] * 9 + [c_int64, c_int, c_float, c_float, c_uint64, c_int, c_int, c_void_p]),
    "vlpet_sublayer_tail_bwd": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_float, c_uint64, c_int, c_int, c_void_p]),
    "vlpet_sublayer_tail_bwd_out": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_float, c_uint64, c_int, c_void_p]),
    "vlpet_sublayer_tail_reduce": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vlpet_layernorm_bwd_xhat": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_void_p]),
    "vlpet_colsum": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vlpet_colsum_partial": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "vlpet_reduce_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vlpet_rmsnorm_fwd": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_float, c_int, c_void_p]),
    "vlpet_rmsnorm_bwd": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p]),
    "vlpet_sublayer_tail_rms_fwd": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_float, c_float, c_uint64, c_int, c_void_p]),
    "vlpet_rmsnorm_tail_bwd": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_float, c_uint64, c_int, c_void_p]),
    "vlpet_attn_fwd": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_attn_bwd": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_attn_fwd_ld": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_attn_bwd_ld": (c_int, [c_void_p] * 10 + [c_int] * 7 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_attn_fwd_bias": (c_int, [c_void_p] * 8 + [c_int] * 7 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_attn_bwd_bias": (c_int, [c_void_p] * 12 + [c_int] * 7 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_concat_dropout_fwd": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_int, c_int, c_float, c_uint64, c_int, c_void_p]),
    "vlpet_concat_dropout_bwd": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_int, c_int, c_float, c_uint64, c_int, c_void_p]),
    "vlpet_attn_fwd_kv": (c_int, [c_void_p] * 8 + [c_int] * 8 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_attn_bwd_kv": (c_int, [c_void_p] * 12 + [c_int] * 8 + [c_float, c_float, c_uint64, c_void_p]),
    "vlpet_ce_loss_fwd": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "vlpet_ce_loss_fwd_checked": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "vlpet_ce_loss_bwd": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "vlpet_act_dropout_fwd": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_float, c_uint64, c_int, c_void_p]),
    "vlpet_act_dropout_bwd": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_float, c_uint64, c_int, c_void_p]),
    "vlpet_lora_delta_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_uint64, c_void_p, c_void_p,
                                     c_void_p, c_int, c_void_p, c_size_t, c_int64, c_int, c_int, c_float, c_int,
                                     c_void_p]),
}


def load():
    """Load the shared library (built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"vl-pet_amd: HIP library not found at {LIB_PATH}; run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (there is no CPU fallback for the PET hot path)")
    # The library shares torch's HIP runtime (same libamdhip64 SONAME).  Loading it registers its code objects,
    # which initialises the runtime; if that happens BEFORE torch has initialised its device context, later launches
    # from this library fail with hipErrorNoDevice (seen with build() followed by smoke() in one process).
    # So: let torch bring the device up first whenever a GPU is present.
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().vlpet_error_string(code)
        raise RuntimeError(f"vl-pet_amd: {what} failed: {msg.decode() if msg else code} (code {code})")
