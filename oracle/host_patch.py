"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): run the host model with every HIP-backed op of the hot
path routed to the CPU restatement in oracle/vlpet_oracle.py.  Used by the parity tests (as the checker) and by
bench.py's ``cpu_baseline`` leg; the product never imports this."""
from __future__ import annotations

import contextlib

import torch.nn.functional as F

from . import vlpet_oracle as O


@contextlib.contextmanager
def cpu_reference_ops():
    import vlpet_amd.host.bart as HB
    from vlpet_amd.adapters.adapter_modeling import Adapter
    from vlpet_amd.visual import Downsample, VisualEmbedding

    def apply_pet(module, which, x1, x2, config):                       # K1 (all four granularity gates)
        downs = getattr(module, f"{which}_adapter_multihead_down")
        up = getattr(module, f"{which}_adapter_multihead_up")
        pre = f"encoder_{which}_adapter_gating_"
        gd, gu = getattr(module, pre + "large_x_down", None), getattr(module, pre + "large_x_up", None)
        small, midx, midy = (getattr(module, pre + k, None) for k in ("small_xy_cat", "middle_xy_add", "middle_ia3_add"))
        if gd is not None:
            mode, gate = O.GATE_LARGE, dict(down_w=gd.weight, down_b=gd.bias, up_w=gu.weight, up_b=gu.bias)
        elif small is not None:
            mode, gate = O.GATE_SMALL, dict(w=small.weight, b=small.bias)
        elif midx is not None:
            mode, gate = O.GATE_MIDDLE_X, dict(w=midx.weight, b=midx.bias)
        elif midy is not None:
            mode, gate = O.GATE_MIDDLE_Y, dict(z=midy)
        else:
            mode, gate = O.GATE_NONE, None
        flag = lambda k, v: float(getattr(config, v)) if getattr(config, k, False) else 1.0
        return O.encoder_adapter_gate(
            x1, x2, [m.weight for m in downs], [m.bias for m in downs], up.weight, up.bias, gate, mode,
            bool(getattr(config, "use_encoder_adapter_gating_add", False)),
            flag("use_encoder_adapter_scaling", "encoder_adapter_scaling_factor"),
            flag("use_encoder_x2_scaling", "encoder_x2_scaling_factor"),
            flag("use_encoder_gating_scaling", "encoder_gating_scaling_factor"))

    def fused(self, x, residual, scale=1.0):                            # K2
        return O.parallel_adapter(x, residual, self.down_sampler.weight, self.down_sampler.bias,
                                  self.up_sampler.weight, self.up_sampler.bias, None if scale == 1.0 else scale)

    def visual(self, feats, pos, img_order_ids=None, obj_order_ids=None):   # K4
        fe, pe = self.feat_embedding, self.absolute_vis_pos_embedding
        return O.visual_embedding(feats, pos, fe[0].weight, fe[0].bias, fe[1].weight, getattr(fe[1], "bias", None),
                                  pe[0].weight, pe[0].bias, pe[1].weight, getattr(pe[1], "bias", None),
                                  self.img_order_embedding.weight, self.obj_order_embedding.weight,
                                  img_order_ids, obj_order_ids, rms=self.rms_norm)

    def lora_forward(self, x, task):                                    # K3
        keep, p = None, self.lora_dropout_p
        if self.training and p > 0.0:
            import torch
            keep = (torch.rand(x.shape) >= p)
        return O.lora_linear(x, self.weight, self.bias, self.lora_As[task], self.lora_Bs[task], self.scaling, keep, p)

    def tail(residual, h, norm, p, training):                           # K5
        hd = F.dropout(h, p=p, training=training)
        if norm is None:                                                # T5: pre-LN stream, plain residual add
            return O.t5_sublayer_tail(residual, hd)
        return O.bart_sublayer_tail(residual, hd, norm.weight, norm.bias, norm.eps)

    def ffn_activation(x, act, p, training):                            # backbone FFN: activation, then dropout
        y = {"gelu": F.gelu, "relu": F.relu, "gelu_new": O.gelu_new}[act](x)
        return F.dropout(y, p=p, training=training)

    def lm_loss(h, weight, labels, bias=None):                          # LM head + CrossEntropyLoss(reduction='none')
        logits = F.linear(h, weight.to(h.dtype))
        if bias is not None:
            logits = logits + bias.to(h.dtype)
        loss = F.cross_entropy(logits.float().view(-1, logits.shape[-1]), labels.view(-1), ignore_index=-100, reduction="none")
        return loss.view(labels.shape), logits

    def attention_core(q, k, v, num_heads, attn_mask, causal, p, training):   # backbone attention: the eager chain
        B, Lq, E = q.shape
        sh = lambda t: t.view(B, -1, num_heads, E // num_heads).transpose(1, 2)
        out = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), attn_mask=attn_mask, is_causal=causal and attn_mask is None,
                                             dropout_p=p if training else 0.0)
        return out.transpose(1, 2).reshape(B, Lq, E)

    def downsample(self, inputs_tuple, out_dtype=None):
        hw = tuple(self.output_size)
        if len(inputs_tuple) == 4:
            y, b, i, o = O.downsample_nlvr(*inputs_tuple, out_hw=hw)
            return (y if out_dtype is None else y.to(out_dtype)), b, i, o
        x, boxes = inputs_tuple
        y = O.downsample(x, hw)
        return (y if out_dtype is None else y.to(out_dtype)), boxes[:, :y.shape[1]]

    import vlpet_amd.train as TR

    class CpuAdamW:
        """clip_grad_norm_ + transformers.AdamW over the trainer's parameter list (oracle restatement)."""

        def __init__(self, flat, lr, max_norm):
            self.flat, self.max_norm = flat, max_norm
            self.steps = [0] * len(flat.params)            # transformers.AdamW: state['step'] per parameter
            self.state = [(p.data.new_zeros(p.shape), p.data.new_zeros(p.shape)) for p in flat.params]

        def step(self, lr):
            # a parameter that got no gradient this step has grad None in the reference (multitask.py:296-297) and is
            # skipped by clip_grad_norm_ (zero contribution) and by AdamW (no decay, no moments, no step count)
            active = self.flat.active() if self.flat.per_task else [True] * len(self.flat.params)
            grads = [p.grad for p, on in zip(self.flat.params, active) if on]
            O.clip_grad_norm(grads, self.max_norm)
            for k, (name, p, (m, v)) in enumerate(zip(self.flat.names, self.flat.params, self.state)):
                if not active[k]:
                    continue
                self.steps[k] += 1
                wd = 0.0 if any(nd in name for nd in TR.NO_DECAY) else 0.01
                O.hf_adamw_step(p.data, p.grad, m, v, self.steps[k], lr, eps=1e-6, weight_decay=wd)
            self.flat.flat.zero_()

    from vlpet_amd.lora.controller import LoRALinearController
    import vlpet_amd.host.t5 as HT
    fuse_saved = (HB.FUSE_RESIDUAL_GRAD, HT.FUSE_RESIDUAL_GRAD)
    HB.FUSE_RESIDUAL_GRAD = HT.FUSE_RESIDUAL_GRAD = False       # plain autograd on the checker path (no kernel-side hand-over)
    saved = (HB.apply_pet, Adapter.fused, VisualEmbedding.forward, HB.sublayer_tail, Downsample.forward,
             TR.CPU_OPTIMIZER_FACTORY, LoRALinearController.forward, HT.apply_pet, HT.sublayer_tail)
    saved_attn = HB.attention_core
    HB.attention_core = attention_core
    saved_act = (HB.ffn_activation, HT.ffn_activation, HB.lm_loss, HT.lm_loss)
    HB.ffn_activation = HT.ffn_activation = ffn_activation
    HB.lm_loss = HT.lm_loss = lm_loss
    HB.apply_pet, Adapter.fused, VisualEmbedding.forward, HB.sublayer_tail, Downsample.forward = \
        apply_pet, fused, visual, tail, downsample
    TR.CPU_OPTIMIZER_FACTORY = CpuAdamW
    LoRALinearController.forward = lora_forward
    HT.apply_pet, HT.sublayer_tail = apply_pet, tail
    try:
        yield
    finally:
        (HB.apply_pet, Adapter.fused, VisualEmbedding.forward, HB.sublayer_tail, Downsample.forward,
         TR.CPU_OPTIMIZER_FACTORY, LoRALinearController.forward, HT.apply_pet, HT.sublayer_tail) = saved
        HB.FUSE_RESIDUAL_GRAD, HT.FUSE_RESIDUAL_GRAD = fuse_saved
        HB.ffn_activation, HT.ffn_activation, HB.lm_loss, HT.lm_loss = saved_act
        HB.attention_core = saved_attn
