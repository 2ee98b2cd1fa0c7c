"""Multitask fine-tuning step around the PET hot path: freeze rules, synthetic batches, loss
reduction, data-parallel gradient exchange, clip + AdamW.

Restated from the reference trainer (host logic only; no code shared):
  * trainable set      trainer_base.py:268-270,308-542  (name-substring rules)
  * init overrides     trainer_base.py:544-599
  * optimizer          trainer_base.py:627-732  (AdamW eps 1e-6, wd 0.01 except bias / LayerNorm.weight,
                       linear warm-up then linear decay)
  * step               multitask.py:217-342  (fwd, bwd, clip_grad_norm 5.0, step, scheduler, grads=None)
  * loss               vqa_model.py:216-227 (mask, per-sample mean, * score, batch mean),
                       caption_model.py / nlvr_model.py (mean over non-ignored tokens)
  * task order         multitask_data.py:34-52 (epoch-seeded shuffle, identical on every rank)
  * batch sizes        multitask.py:682-695

Data parallelism (the reference wraps DDP but bypasses DDP.forward, multitask.py:231-239, so it never
arms the reducer): here the ~6 M trainable gradients live in ONE flat fp32
buffer, every parameter's ``.grad`` is a view into it, and the buffer is all-reduced over RCCL in a few
buckets ordered by backward readiness (decoder adapters -> encoder layers 5..0 -> visual embedding),
each launched asynchronously from a post-accumulate hook so it overlaps the remaining frozen backward.
No per-step barrier.
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

TASK_BATCH = {"vqa": lambda b: b, "gqa": lambda b: int(b * 100 / 60), "nlvr": lambda b: int(b * 20 / 60),
              "caption": lambda b: int(b * 50 / 60)}
TEXT_LEN = {"vqa": 20, "gqa": 20, "nlvr": 20, "caption": 40}
TARGET_LEN = {"vqa": 5, "gqa": 5, "nlvr": 2, "caption": 20}


# ------------------------------------------------------------------ trainable set
def trainable_names(model: nn.Module, config) -> List[str]:
    """Apply the reference's substring rules; returns the names left trainable."""
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = False
    any_adapter = bool(config.use_encoder_adapter_down_multihead or
                       config.use_decoder_enc_attn_value_parallel_adapter_down_dim)
    any_gate = bool(config.use_encoder_adapter_gating_large_x_lowrank or
                    config.use_encoder_adapter_gating_small_xy_cat or
                    config.use_encoder_adapter_gating_middle_xy_add or
                    config.use_encoder_adapter_gating_middle_ia3_add)
    for n, p in model.named_parameters():
        on = False
        if not config.freeze_vis_emb and "visual_embedding" in n:
            on = True
        if config.use_lora and ("lora" in n or "bias" in n):
            on = True
        if config.unfreeze_encoder_layer_norms and "encoder." in n and ("layer_norm" in n or "layernorm" in n):
            on = True
        if any_gate and "gating" in n:
            on = True
        if any_adapter and "adapter" in n:
            on = True
        if on:
            p.requires_grad = True
            names.append(n)
    return names


def weight_initialization(model: nn.Module, config):
    with torch.no_grad():
        for n, p in model.named_parameters():
            if config.use_encoder_multihead_up_zero_init and "adapter_multihead_up" in n:
                p.zero_()
            if config.use_encoder_gating_large_x_lowrank_up_zero_init and "adapter_gating_large_x_up" in n:
                p.zero_()
            if config.use_decoder_enc_vpa_up_zero_init and "attn_value_parallel_adapter" in n and "up_sampler" in n:
                p.zero_()
            # trainer_base.py:577-599
            if getattr(config, "use_encoder_gating_small_up_zero_init", False) and "adapter_gating_small_xy_cat" in n:
                p.zero_()
            if getattr(config, "use_encoder_gating_middle_up_zero_init", False) and "adapter_gating_middle_xy_add" in n:
                p.zero_()
            if getattr(config, "use_encoder_gating_middle_ia3_one_init", False) and "adapter_gating_middle_ia3_add" in n:
                p.fill_(1.0)
            if getattr(config, "use_encoder_gating_middle_ia3_zero_init", False) and "adapter_gating_middle_ia3_add" in n:
                p.zero_()


def cast_frozen(model: nn.Module, dtype: torch.dtype):
    """Frozen backbone weights live in the compute dtype; trainable parameters stay fp32 masters
    (the pack kernel and HostLayerNorm cast them on use)."""
    for p in model.parameters():
        if not p.requires_grad and p.is_floating_point():
            p.data = p.data.to(dtype)
    for b in model.buffers():
        if b.is_floating_point():
            b.data = b.data.to(dtype)


# ------------------------------------------------------------------ synthetic data
def synthetic_batch(task: str, batch: int, config, device, gen: torch.Generator, feat_dtype=torch.float32):
    """Synthetic CLIP-feature + token batch with the reference loaders' shapes:
    RN101 grid 7x7 = 49 features of 2048, zero boxes, fixed-length text, short targets."""
    V = config.vocab_size - 200
    L, T = TEXT_LEN[task], TARGET_LEN[task]
    n_grid = 49
    ids = torch.randint(5, V, (batch, L), device=device, generator=gen)
    labels = torch.randint(5, V, (batch, T), device=device, generator=gen)
    if task == "nlvr":
        feats = torch.randn(batch, 2 * n_grid, int(config.feat_dim), device=device, generator=gen, dtype=feat_dtype)
        boxes = torch.zeros(batch, 2 * n_grid, 4, device=device, dtype=feat_dtype)
        img = torch.cat([torch.zeros(n_grid, dtype=torch.long), torch.ones(n_grid, dtype=torch.long)])
        obj = torch.cat([torch.arange(n_grid), torch.arange(n_grid)])
        vis = (feats, boxes, img.to(device).unsqueeze(0).expand(batch, -1), obj.to(device).unsqueeze(0).expand(batch, -1))
    else:
        feats = torch.randn(batch, n_grid, int(config.feat_dim), device=device, generator=gen, dtype=feat_dtype)
        boxes = torch.zeros(batch, n_grid, 4, device=device, dtype=feat_dtype)
        vis = (feats, boxes)
    return dict(task=task, input_ids=ids, vis_inputs=vis, labels=labels,
                scores=torch.ones(batch, device=device))


def epoch_task_order(tasks: Sequence[str], steps_per_task: Dict[str, int], epoch: int) -> List[str]:
    order = []
    for t in tasks:
        order += [t] * steps_per_task[t]
    random.Random(epoch).shuffle(order)
    return order


def task_loss(per_token: torch.Tensor, labels: torch.Tensor, scores: Optional[torch.Tensor], task: str):
    mask = (labels != -100).float()
    if task in ("vqa", "gqa"):
        loss = (per_token * mask).sum(1) / mask.sum(1).clamp(min=1)
        if scores is not None:
            loss = loss * scores
        return loss.mean()
    return (per_token * mask).sum() / mask.sum().clamp(min=1)


# ------------------------------------------------------------------ flat gradients + DP exchange
class FlatGrads:
    def __init__(self, model: nn.Module, world_size: int = 1, n_buckets: int = 3, process_group=None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]

        def readiness(n):          # backward produces decoder grads first, visual embedding last
            if ".decoder." in n:
                return 0
            if "visual_embedding" in n or "layernorm_embedding" in n:
                return 2
            return 1
        named.sort(key=lambda t: readiness(t[0]))
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.world_size = world_size
        self.group = process_group
        off = 0
        self.slices = []
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.slices.append((off, off + n))
            off += n
        # buckets = contiguous ranges of roughly equal size, cut at parameter boundaries
        self.bucket_of = []
        self.buckets = []
        n_buckets = max(1, min(n_buckets, len(self.params)))
        target = total / n_buckets
        start, cur = 0, 0
        for i, (a, b) in enumerate(self.slices):
            self.bucket_of.append(len(self.buckets))
            cur = b
            if cur - start >= target and len(self.buckets) < n_buckets - 1:
                self.buckets.append((start, cur))
                start = cur
        self.buckets.append((start, total))
        self.bucket_count = [0] * len(self.buckets)
        for bidx in self.bucket_of:
            self.bucket_count[bidx] += 1
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        if world_size > 1:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self._pending[b] += 1
            if self._pending[b] == self.bucket_count[b]:
                self._launch(b)
        return hook

    def _launch(self, b):
        a, e = self.buckets[b]
        self._handles.append(dist.all_reduce(self.flat[a:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, used: Optional[Sequence[bool]] = None):
        """Wait for the bucket all-reduces (launching any bucket whose parameters did not all receive a
        gradient this step -- per-task adapters / LoRA leave other tasks' grads at zero) and average."""
        if self.world_size > 1:
            for b in range(len(self.buckets)):
                if self._pending[b] != self.bucket_count[b]:
                    self._launch(b)
            for h in self._handles:
                h.wait()
            self._handles.clear()
            self._pending = [0] * len(self.buckets)
            self.flat.div_(self.world_size)

    def zero(self):
        self.flat.zero_()
        for p, (a, b) in zip(self.params, self.slices):     # re-attach in case something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat[a:b].data_ptr():
                p.grad = self.flat[a:b].view_as(p)

    def clip_(self, max_norm: float) -> torch.Tensor:
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


def build_optimizer(model: nn.Module, lr=1e-3, weight_decay=0.01, eps=1e-6):
    no_decay = ("bias", "LayerNorm.weight")
    decay, nodecay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (nodecay if any(nd in n for nd in no_decay) else decay).append(p)
    groups = [dict(params=decay, weight_decay=weight_decay), dict(params=nodecay, weight_decay=0.0)]
    return torch.optim.AdamW(groups, lr=lr, eps=eps, fused=decay[0].is_cuda if decay else False)


def linear_schedule(optimizer, warmup_steps: int, total_steps: int):
    def f(step):
        if step < warmup_steps:
            return step / max(1, warmup_steps)
        return max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, f)


class Trainer:
    """One process per GPU.  ``step(batch)`` = forward, backward (with overlapped gradient exchange),
    clip, AdamW, scheduler."""

    def __init__(self, model: nn.Module, config, lr=1e-3, clip=5.0, total_steps=1000, warmup_ratio=0.1,
                 world_size=1, n_buckets=3, process_group=None):
        self.model, self.config, self.clip = model, config, clip
        self.flat = FlatGrads(model, world_size, n_buckets, process_group)
        self.optim = build_optimizer(model, lr=lr)
        self.sched = linear_schedule(self.optim, int(total_steps * warmup_ratio), total_steps)

    def step(self, batch) -> torch.Tensor:
        self.flat.zero()
        per_token, _ = self.model(batch["input_ids"], batch["vis_inputs"], batch["labels"], batch["task"])
        loss = task_loss(per_token, batch["labels"], batch.get("scores"), batch["task"])
        loss.backward()
        self.flat.finish()
        self.flat.clip_(self.clip)
        self.optim.step()
        self.sched.step()
        return loss.detach()
