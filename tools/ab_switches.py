"""A/B switches of the host package as module attributes, set from VLPET_* environment variables by the tools (and bench.py when
VLPET_AB=1): the product package itself reads nothing from the environment at import.
    VLPET_NO_LINK=1, VLPET_NO_GEMM_LINK=1, VLPET_NO_NORM_LINK=1, VLPET_NO_LORA_LINK=1, VLPET_NO_BIAS_GRAD_KERNEL=1, VLPET_NO_FUSED_QKV=1,
    VLPET_EAGER_FFN_ACT=1, VLPET_EAGER_LM_LOSS=1, VLPET_EAGER_ATTENTION=1, VLPET_EAGER_RMS_NORM=1, VLPET_SPLIT_WIDE=1, VLPET_SDPA=flash|efficient|math, VLPET_NO_DEFER_REDUCES=1,
    VLPET_K4_FORM=gemm|library|fused, VLPET_SAVE_PRENORM=auto|0|1, VLPET_NO_TAIL_NORM_FUSION=1, VLPET_NO_ALIAS_RESIDUAL_GRAD=1,
    VLPET_K1_BWD_FROM_X2=1, VLPET_DEFER_FINALIZE=1, VLPET_FINALIZE_SIDE_STREAM=1, VLPET_FINALIZE_LAUNCH=1, VLPET_NO_POS_KERNEL=1, VLPET_NO_FANOUT_SUM=1, VLPET_NO_FUSED_CROSS_KEYS=1, VLPET_NO_CONCAT_DROPOUT=1"""
import os


def apply():
    import vlpet_amd.encoder_pet as EP
    import vlpet_amd.visual as V
    import vlpet_amd.host.bart as HB
    import vlpet_amd.host.t5 as HT
    import vlpet_amd.lora.controller as LC
    on = lambda n: os.environ.get(n, "0") == "1"
    changed = {}

    def put(mod, attr, val):
        if getattr(mod, attr) != val:
            setattr(mod, attr, val)
            changed[f"{mod.__name__}.{attr}"] = val
    put(EP, "SPLIT_WIDE_BOTTLENECK", on("VLPET_SPLIT_WIDE"))
    put(V, "EAGER_RMS_NORM", on("VLPET_EAGER_RMS_NORM"))
    for mod in (HB, HT):
        put(mod, "FUSE_RESIDUAL_GRAD", not on("VLPET_NO_LINK"))
    put(HB, "FUSE_GEMM_GRAD", not on("VLPET_NO_GEMM_LINK"))
    put(HT, "FUSE_NORM_GRAD", not on("VLPET_NO_NORM_LINK"))
    put(HB, "EAGER_FFN_ACT", on("VLPET_EAGER_FFN_ACT"))
    put(HB, "EAGER_LM_LOSS", on("VLPET_EAGER_LM_LOSS"))
    put(HB, "FUSE_BIAS_GRAD", not on("VLPET_NO_BIAS_GRAD_KERNEL"))
    put(HB, "EAGER_ATTENTION", on("VLPET_EAGER_ATTENTION"))
    put(HT, "EAGER_ATTENTION", on("VLPET_EAGER_ATTENTION"))
    put(HB, "FUSE_QKV", not on("VLPET_NO_FUSED_QKV"))
    put(HB, "FUSE_CROSS_KEYS", not on("VLPET_NO_FUSED_CROSS_KEYS")); put(HT, "FUSE_CROSS_KEYS", not on("VLPET_NO_FUSED_CROSS_KEYS"))        # the decoder layers' cross-attention key projections as one GEMM each way
    put(HT, "FUSE_QKV", not on("VLPET_NO_FUSED_QKV"))
    put(HB, "SDPA_BACKEND", os.environ.get("VLPET_SDPA") or None)
    put(LC, "LINK_DELTA_GRAD", not on("VLPET_NO_LORA_LINK"))
    import vlpet_amd.train as TR
    put(TR, "DEFER_PARAM_REDUCES", not on("VLPET_NO_DEFER_REDUCES"))
    put(HT, "FUSE_TAIL_NORM", not on("VLPET_NO_TAIL_NORM_FUSION"))             # T5: tail + the next sublayer's RMS norm as one launch each way
    import vlpet_amd.visproj as VP
    import vlpet_amd.tail as TL
    put(VP, "K4_FORM", os.environ.get("VLPET_K4_FORM") or "gemm")           # gemm (default) | library | fused
    put(VP, "FUSE_POS_BRANCH", not on("VLPET_NO_POS_KERNEL"))                # K4's position / order branch: csrc/vispos.hip (default) | library ops
    pn = os.environ.get("VLPET_SAVE_PRENORM")
    put(TL, "ALIAS_RESIDUAL_GRAD", not on("VLPET_NO_ALIAS_RESIDUAL_GRAD"))        # plain residual tail: d/dx1 = dout handed on without a copy
    put(TL, "SAVE_PRENORM", None if pn in (None, "", "auto") else pn == "1")  # K5: auto (default) | 1 = exact form | 0 = from the output
    import vlpet_amd.functional as VF
    put(VF, "DEFER_FINALIZE", on("VLPET_DEFER_FINALIZE"))                       # weight-gradient finalize launches queued and issued 16 per launch at the end of the backward (default off: no gain)
    put(VF, "FINALIZE_SIDE_STREAM", on("VLPET_FINALIZE_SIDE_STREAM"))          # captured steps: K1's finalize launch as a parallel branch of the graph (default off: slower)
    put(VF, "K1_BWD_FROM_OUTPUT", not on("VLPET_K1_BWD_FROM_X2"))               # gated K1 backward from the forward's output y (default) | from x2
    import vlpet_amd.act as ACT
    put(ACT, "FUSE_CONCAT_DROPOUT", not on("VLPET_NO_CONCAT_DROPOUT"))          # the joint encoder's cat + dropout as one pass each way | torch.cat + F.dropout
    put(VF, "FANOUT_SUM", not on("VLPET_NO_FANOUT_SUM"))                        # the encoder output's gradient summed in one launch (vlpet_sum_n) | autograd's pairwise adds
    put(TR, "IN_LAUNCH_REDUCE", not on("VLPET_FINALIZE_LAUNCH"))               # VLPET_FINALIZE_LAUNCH=1: the K1 / K2 / K3 backward passes end in a finalize launch (round 3) instead of the in-launch reduce-scatter
    return changed
