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

import collections
import math
import os
import random
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

TASK_BATCH = {"vqa": lambda b: b, "gqa": lambda b: int(b * 100 / 60), "nlvr": lambda b: int(b * 20 / 60),
              "caption": lambda b: int(b * 50 / 60),
              # video-text (multitask_video.py:760-860: every task loader gets args.batch_size)
              "tvqa": lambda b: b, "how2qa": lambda b: b, "tvc": lambda b: b, "yc2c": lambda b: b}
# video: subtitles + question + prompt, truncated at 600 tokens (video/tvqa_data.py:211, how2qa_data.py:203, tvc_data.py:210,
# yc2c_data.py:206); targets capped at 20 (tvqa_data.py:231), QA answers are a few tokens
TEXT_LEN = {"vqa": 20, "gqa": 20, "nlvr": 20, "caption": 40, "tvqa": 600, "how2qa": 600, "tvc": 600, "yc2c": 600}
TARGET_LEN = {"vqa": 5, "gqa": 5, "nlvr": 2, "caption": 20, "tvqa": 3, "how2qa": 3, "tvc": 20, "yc2c": 20}
VIDEO_TASKS = ("tvqa", "how2qa", "tvc", "yc2c")
VIDEO_FRAMES = 64       # clip-vit frame embeddings resized to n_boxes = 64 (video/how2qa_data.py:34-44,163)


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
def synthetic_batch(task: str, batch: int, config, device, gen: torch.Generator, feat_dtype=torch.float32, no_padding: bool = True):
    """Synthetic CLIP-feature + token batch with the reference loaders' shapes:
    RN101 grid 7x7 = 49 features of 2048, zero boxes, fixed-length text, short targets."""
    V = config.vocab_size - 200
    L, T = TEXT_LEN[task], TARGET_LEN[task]
    n_grid = 49
    ids = torch.randint(5, V, (batch, L), device=device, generator=gen)
    labels = torch.randint(5, V, (batch, T), device=device, generator=gen)
    if task in VIDEO_TASKS:
        # [B, 64, feat_dim = 512] frame features, zero boxes (multitask_video.py:738; video/video_model.py:34-36)
        feats = torch.randn(batch, VIDEO_FRAMES, int(config.feat_dim), device=device, generator=gen, dtype=feat_dtype)
        boxes = torch.zeros(batch, VIDEO_FRAMES, 4, device=device, dtype=feat_dtype)
        return dict(task=task, input_ids=ids, vis_inputs=(feats, boxes), labels=labels, scores=None, no_padding=no_padding)
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
    # ids are drawn from [5, V): no pad token (id 1) occurs, every row has the full length.  no_padding = True tells the host
    # that (it then skips building the all-ones input_ids.ne(pad) mask); False = the reference's default path, which builds
    # and applies the mask every step (src/modeling_bart.py:817-818, 995-996)
    return dict(task=task, input_ids=ids, vis_inputs=vis, labels=labels,
                scores=torch.ones(batch, device=device), no_padding=no_padding)


def epoch_task_order(tasks: Sequence[str], steps_per_task: Dict[str, int], epoch: int) -> List[str]:
    order = []
    for t in tasks:
        order += [t] * steps_per_task[t]
    random.Random(epoch).shuffle(order)
    return order


def task_loss(per_token: torch.Tensor, labels: torch.Tensor, scores: Optional[torch.Tensor], task: str):
    mask = (labels != -100).float()
    if task in ("vqa", "gqa") or task in VIDEO_TASKS:      # video/video_model.py:77-87: per-sample mean, then batch mean
        loss = (per_token * mask).sum(1) / mask.sum(1).clamp(min=1)
        if scores is not None and task not in VIDEO_TASKS:      # the video heads do not weight by score (:85)
            loss = loss * scores
        return loss.mean()
    return (per_token * mask).sum() / mask.sum().clamp(min=1)


# ------------------------------------------------------------------ flat trainable state + DP exchange
class GradSink:
    """Destination of a weight gradient inside the flat gradient buffer.  The PET autograd functions hand
    ``view`` straight to the weight-gradient kernels (no temporary, no AccumulateGrad add kernel) and then call
    ``done()``, which does the bucket accounting the post-accumulate hook would have done."""

    def __init__(self, owner: "FlatGrads", view: torch.Tensor, indices: Sequence[int]):
        self.owner, self.view, self.indices = owner, view, tuple(indices)
        self.epoch = -1

    def take(self) -> Optional[torch.Tensor]:
        """The view if this is the first write of the step (the kernels overwrite), else None (-> autograd adds).

        Under data parallelism a sunk parameter must be used ONCE per backward: ``done()`` of the first use already
        counted it ready, so its bucket may be in flight when a second contribution arrives.  (True for every PET
        parameter of the reference: one adapter / gate / LoRA pair per module and step.)"""
        if not self.owner.sinks_enabled:
            return None
        if self.epoch == self.owner.epoch:
            from . import functional as VF
            VF.flush_reduces()          # a queued first write (deferred finalize / reduction) must land before autograd adds the second
            if self.owner.dp:
                raise RuntimeError("vl-pet_amd: a parameter with a direct-write gradient slot was used twice in one "
                                   "backward under data parallelism; build FlatGrads(sinks=False) for such models")
            return None
        self.epoch = self.owner.epoch
        return self.view

    def done(self):
        for i in self.indices:
            self.owner._ready(i)


def _flat_order(named):
    """Backward-readiness classes (decoder first, visual embedding last); inside a class the parameters keep
    their order except that the N_h blocks of a multi-head down projection become adjacent
    ([w0..w3 | b0..b3]) so that their gradient is ONE [r, d] block of the flat buffer."""
    import re

    def readiness(n):
        if ".decoder." in n:
            return 0
        if "visual_embedding" in n or "layernorm_embedding" in n:
            return 2
        return 1
    pat = re.compile(r"(.*adapter_multihead_down)\.(\d+)\.(weight|bias)$")
    first_seen = {}
    keyed = []
    for pos, (n, p) in enumerate(named):
        mm = pat.match(n)
        if mm:
            g = mm.group(1)
            first_seen.setdefault(g, pos)
            keyed.append(((readiness(n), first_seen[g], 0 if mm.group(3) == "weight" else 1, int(mm.group(2))), n, p))
        else:
            keyed.append(((readiness(n), pos, 0, 0), n, p))
    keyed.sort(key=lambda t: t[0])
    return [(n, p) for _, n, p in keyed]


def _has_per_task_params(names, params) -> bool:
    """True when some trainable parameter belongs to ONE task (``...adapters.<task>.*`` / ``lora_As.<task>`` entries that
    are distinct tensors per task).  Shared modules registered under every task key (use_single_adapter /
    use_single_lora, adapters/adapter_controller.py:49-58, lora/controller.py:72-80) are one tensor with one name."""
    import re
    pat = re.compile(r"(.*\.(?:adapters|lora_As|lora_Bs))\.([^.]+)(\..*)?$")
    seen = {}
    for n in names:
        m = pat.match(n)
        if m:
            seen.setdefault((m.group(1), m.group(3) or ""), set()).add(m.group(2))
    return any(len(v) > 1 for v in seen.values())


class FlatGrads:
    """Flat fp32 gradient buffer with ``.grad`` views, bucketed asynchronous all-reduce, and (optionally) the
    parameters themselves + Adam moments as flat buffers for the fused optimizer."""

    def __init__(self, model: nn.Module, world_size: int = 1, n_buckets: int = 3, process_group=None,
                 flatten_params: bool = False, sinks: bool = False, force_collectives: bool = False):
        named = _flat_order([(n, p) for n, p in model.named_parameters() if p.requires_grad])
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        total = sum(p.numel() for p in self.params)
        total_pad = (total + 3) // 4 * 4
        dev = self.params[0].device
        self.total = total
        self.flat = torch.zeros(total_pad, dtype=torch.float32, device=dev)
        self.world_size = world_size
        self.dp = world_size > 1 or force_collectives      # force: run the bucket all-reduces on a 1-rank group (RCCL smoke test)
        self.group = process_group
        self.epoch = 0
        self.defer = False          # True: gradients are exchanged by finish() only (no bucket launches from the backward: graph capture)
        self.sinks_enabled = bool(sinks)
        off = 0
        self.slices = []
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.slices.append((off, off + n))
            off += n
        self.flat_p = None
        self.flat_p_io = None       # IO-dtype shadow of flat_p (refresh_io_shadow): the trainable biases' bf16 copies as views of ONE buffer
        self.io_epoch = -1
        if flatten_params:
            self.flat_p = torch.zeros(total_pad, dtype=torch.float32, device=dev)
            for p, (a, b) in zip(self.params, self.slices):
                self.flat_p[a:b].copy_(p.data.reshape(-1).float())
                p.data = self.flat_p[a:b].view_as(p)
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
        self._ready_epoch = [-1] * len(self.params)
        self._handles = []
        self._hooks = []
        # Per-task parameters (use_single_adapter / use_single_lora off): only the current task's adapters / LoRA matrices
        # get a gradient in a step, and the optimizer must know which (transformers.AdamW skips grad-None parameters).
        self.per_task = _has_per_task_params(self.names, self.params)
        if self.dp or self.per_task:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        if sinks:
            self._attach_sinks()

    # ---- direct-write destinations for the weight-gradient kernels
    def _attach_sinks(self):
        import re
        pat = re.compile(r"(.*adapter_multihead_down)\.(\d+)\.(weight|bias)$")
        groups: Dict[tuple, List[int]] = {}
        for i, n in enumerate(self.names):
            mm = pat.match(n)
            if mm:
                groups.setdefault((mm.group(1), mm.group(3)), []).append(i)
        in_group = set()
        for (_, kind), idx in groups.items():
            idx.sort()
            a, b = self.slices[idx[0]][0], self.slices[idx[-1]][1]
            contiguous = all(self.slices[idx[k]][1] == self.slices[idx[k + 1]][0] for k in range(len(idx) - 1))
            if not contiguous:
                continue
            first = self.params[idx[0]]
            rows = sum(self.params[i].shape[0] for i in idx)
            view = self.flat[a:b].view(rows, *first.shape[1:]) if kind == "weight" else self.flat[a:b]
            first._vlpet_block_sink = GradSink(self, view, idx)
            in_group.update(idx)
        for i, p in enumerate(self.params):
            if i not in in_group:
                a, b = self.slices[i]
                p._vlpet_sink = GradSink(self, self.flat[a:b].view_as(p), [i])

    def _ready(self, i):
        """Parameter i has its gradient of this step in the flat buffer.  Idempotent per step: a sunk parameter is
        reported by ``GradSink.done()`` AND by its post-accumulate hook (autograd runs the hook even though the PET
        function returned None for it) -- counting both launched the bucket's all-reduce before the other members had
        their gradients (caught by tests/test_dp_gloo.py and tests/test_gpu_dp.py)."""
        if self._ready_epoch[i] == self.epoch:
            return
        self._ready_epoch[i] = self.epoch
        if self.dp and not self.defer:
            b = self.bucket_of[i]
            self._pending[b] += 1
            if self._pending[b] == self.bucket_count[b]:
                self._launch(b)

    def active(self):
        """Per parameter: did it receive a gradient in the current step (meaningful when hooks are registered)."""
        return [e == self.epoch for e in self._ready_epoch]

    def _make_hook(self, i):
        def hook(param):
            self._ready(i)
        return hook

    def _launch(self, b):
        a, e = self.buckets[b]
        self._join_side_stream()
        self._handles.append(dist.all_reduce(self.flat[a:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _join_side_stream(self):
        """Weight gradients written on the side stream (functional.WGRAD_STREAM) must be visible to whatever reads
        the flat buffer next (a bucket all-reduce, the optimizer)."""
        from . import functional as VF
        if VF.WGRAD_STREAM is not None and self.flat.is_cuda:
            torch.cuda.current_stream().wait_stream(VF.WGRAD_STREAM)

    def finish(self, average: bool = True):
        """Wait for the bucket all-reduces (launching any bucket whose parameters did not all receive a
        gradient this step -- per-task adapters / LoRA leave other tasks' grads at zero) and average
        (``average=False``: the caller folds 1/world_size into its optimizer kernel)."""
        self._join_side_stream()
        if self.dp:
            for b in range(len(self.buckets)):
                if self._pending[b] < self.bucket_count[b]:
                    self._launch(b)
            for h in self._handles:
                h.wait()
            self._handles.clear()
            self._pending = [0] * len(self.buckets)
            if average:
                self.flat.div_(self.world_size)

    def refresh_io_shadow(self, dtype=torch.bfloat16):
        """One cast launch over the flat parameter buffer after an optimizer step instead of one ``b.to(bf16)`` per trainable
        bias and forward (96 per step in the LoRA runs, which train every bias next to frozen bf16 weights; under graph replay
        they were 96 nodes).  ``functional.io_view(p, dtype)`` hands out the parameter's slice while the shadow is current."""
        from . import functional as VF
        if self.flat_p is None or not self.flat_p.is_cuda:
            return
        if self.flat_p_io is None or self.flat_p_io.dtype != dtype:
            self.flat_p_io = torch.empty_like(self.flat_p, dtype=dtype)
            for p, (a, b) in zip(self.params, self.slices):
                p._vlpet_io = (self, self.flat_p_io[a:b].view_as(p))
        self.flat_p_io.copy_(self.flat_p)
        self.io_epoch = VF.WEIGHTS_EPOCH

    def begin_step(self, zero: bool = True):
        """New accumulation epoch: every sink accepts one direct write again."""
        self.epoch += 1
        if zero:
            self.flat.zero_()
        for p, (a, b) in zip(self.params, self.slices):     # re-attach in case something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat[a:b].data_ptr():
                p.grad = self.flat[a:b].view_as(p)

    def zero(self):
        self.begin_step(zero=True)

    def clip_(self, max_norm: float) -> torch.Tensor:
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


NO_DECAY = ("bias", "LayerNorm.weight")     # trainer_base.py:635 (substring rule; BART's *_layer_norm.weight DOES decay)


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")


def use_tuned_gemms(path: Optional[str] = None, tune: bool = False) -> bool:
    """Library-GEMM solution selection for the frozen backbone (the q/k/v/o, FFN and LM-head products stay plain
    hipBLASLt / rocBLAS GEMMs): PyTorch's TunableOp with the table measured on an MI355X for the shapes of the
    benchmark's four task batches (``tuning/tunableop_gfx950.csv``; 22,017 -> 22,566 samples/s on the same box).
    ``tune=True`` measures missing shapes on first use and appends them (minutes per run).  The table carries the
    library versions it was measured with; TunableOp ignores it when they differ.  Returns whether it is active."""
    import torch.cuda.tunable as tunable
    path = path or TUNED_GEMMS
    if not tune and not os.path.exists(path):
        return False
    tunable.set_filename(path, insert_device_ordinal=False)     # the same table for every rank of a node
    tunable.enable(True)
    tunable.tuning_enable(bool(tune))
    return True


def tail_recheck_every() -> int:
    from . import tail as _tail
    return max(1, int(_tail.PRENORM_RECHECK) - 1)       # (one step ahead of the per-LayerNorm expiry, so that one never triggers under a trainer)


def lr_at(step: int, base_lr: float, warmup_steps: int, total_steps: int) -> float:
    """get_linear_schedule_with_warmup (trainer_base.py:633-720): factor for the update with 0-based index ``step``."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    return base_lr * max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))


class FusedAdamW:
    """Global-norm clip + AdamW (transformers.optimization.AdamW semantics, trainer_base.py:690-701) over the
    flat trainable buffer: two HIP launches per step (csrc/optim.hip).  GPU only."""

    def __init__(self, flat: FlatGrads, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, max_norm=5.0,
                 variant=0):
        from . import _lib
        if flat.flat_p is None or not flat.flat.is_cuda:
            raise RuntimeError("vl-pet_amd: FusedAdamW needs FlatGrads(flatten_params=True) on the GPU")
        self.lib = _lib.load()
        self.flat, self.lr, self.betas, self.eps, self.wd, self.max_norm, self.variant = \
            flat, lr, betas, eps, weight_decay, max_norm, variant
        dev = flat.flat.device
        n = flat.flat.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        mask = torch.zeros(n, dtype=torch.uint8)
        for name, (a, b) in zip(flat.names, flat.slices):
            if not any(nd in name for nd in NO_DECAY):
                mask[a:b] = 1
        self.decay = mask.to(dev)
        self.repack_every_pair = False      # set by a trainer that replays captured steps (no Python forward marks the pairs as used)
        # bf16 shadow of the trainable parameters, refreshed after every step: only where it is used (trainable biases of frozen
        # bf16 projections = the LoRA runs; FlatGrads.refresh_io_shadow)
        self.io_shadow = any(n.endswith(".bias") and p.dim() == 1 for n, p in zip(flat.names, flat.params)) and not flat.per_task
        if self.io_shadow:
            flat.refresh_io_shadow()
        self.nb = self.lib.vlpet_optim_blocks(n)
        self.partials = torch.empty(self.nb, dtype=torch.float32, device=dev)
        self.norm = torch.zeros((), dtype=torch.float32, device=dev)
        self.t = 0
        # per-parameter step counts (transformers.AdamW state['step']; a parameter without a gradient is skipped): only
        # needed when some parameters belong to one task -- otherwise every parameter is updated at every step
        self.sliced = bool(flat.per_task)
        if self.sliced:
            so = torch.zeros(n, dtype=torch.int32)
            for k, (a, b) in enumerate(flat.slices):
                so[a:b] = k
            self.slice_of = so.to(dev)
            self.steps = [0] * len(flat.slices)
            # the per-parameter bias corrections travel host -> device every step through a small ring of pinned buffers: the
            # copy is asynchronous and the trainer never syncs with the GPU, so a single staging buffer could be rewritten by the
            # host (one step ahead) before the previous step's DMA had read it; each slot's event says when it is free again
            self._bc_ring = [torch.zeros(len(flat.slices), 2, dtype=torch.float32).pin_memory() for _ in range(4)]
            self._bc_free = [None] * len(self._bc_ring)
            self.bc_dev = torch.zeros(len(flat.slices), 2, dtype=torch.float32, device=dev)

    def step(self, lr: Optional[float] = None):
        from . import _lib
        from . import functional as VF
        f = self.flat
        n = f.flat.numel()
        st = torch.cuda.current_stream().cuda_stream
        self.t += 1
        rc = self.lib.vlpet_grad_sumsq(f.flat.data_ptr(), n, self.partials.data_ptr(), st)
        _lib.check(rc, "vlpet_grad_sumsq")
        if self.sliced:
            b1, b2 = self.betas
            slot = self.t % len(self._bc_ring)
            if self._bc_free[slot] is not None:
                self._bc_free[slot].synchronize()       # (only ever waits when the host is a whole ring ahead of the GPU)
            bc_host = self._bc_ring[slot]
            for k, on in enumerate(f.active()):
                if on:
                    self.steps[k] += 1
                    bc_host[k, 0] = 1.0 - b1 ** self.steps[k]
                    bc_host[k, 1] = math.sqrt(1.0 - b2 ** self.steps[k])
                else:
                    bc_host[k, 0] = -1.0
            self.bc_dev.copy_(bc_host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._bc_free[slot] = ev
            rc = self.lib.vlpet_adamw_step_sliced(
                f.flat_p.data_ptr(), f.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.decay.data_ptr(), n,
                self.partials.data_ptr(), self.nb, float(self.max_norm), 1.0 / f.world_size,
                float(self.lr if lr is None else lr), b1, b2, self.eps, self.wd, self.slice_of.data_ptr(),
                self.bc_dev.data_ptr(), self.variant, 1, self.norm.data_ptr(), st)
            _lib.check(rc, "vlpet_adamw_step_sliced")
            VF.bump_weights_epoch()
            VF.repack_all(self.repack_every_pair)        # the adapters' fragment packs for the next step: a few batched launches
            if self.io_shadow:
                f.refresh_io_shadow()
            return
        rc = self.lib.vlpet_adamw_step(f.flat_p.data_ptr(), f.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                       self.decay.data_ptr(), n, self.partials.data_ptr(), self.nb, float(self.max_norm),
                                       1.0 / f.world_size, float(self.lr if lr is None else lr), self.betas[0],
                                       self.betas[1], self.eps, self.wd, self.t, self.variant, 1, self.norm.data_ptr(), st)
        _lib.check(rc, "vlpet_adamw_step")
        VF.bump_weights_epoch()
        VF.repack_all(self.repack_every_pair)            # the adapters' fragment packs for the next step: a few batched launches
        if self.io_shadow:
            f.refresh_io_shadow()


# Set by the test / CPU-baseline harness: factory(flat: FlatGrads, lr, max_norm) -> object with .step(lr).
# The product has no CPU optimizer of its own.
CPU_OPTIMIZER_FACTORY = None


_REGISTERED_SEED_CTR = None     # address of the device step counter currently registered with the library (Trainer.enable_graph / close)
MAX_GRAPHS = 16              # captured step shapes a trainer keeps (least recently replayed evicted); ragged data beyond it runs eager + capture
LABEL_CHECK_EVERY = 100      # steps between reads of the device-side bad-label tally (0: never)
IN_LAUNCH_REDUCE = True         # A/B switch (tools/ab_switches.py): False = the K1 / K2 / K3 backward passes always end in a finalize launch (round 3)
DEFER_PARAM_REDUCES = True      # A/B switch (tools/ab_switches.py): False = one reduce launch per parameter gradient, as before round 4


class Trainer:
    """One process per GPU.  ``step(batch)`` = forward, backward (with overlapped gradient exchange),
    clip, AdamW, scheduler -- multitask.py:217-342.

    ``graph=True`` (GPU only): forward + loss + backward of a step are captured ONCE per batch signature (task, shapes) with
    hipGraph (``torch.cuda.graph``) and replayed from then on; gradient exchange, clip and AdamW stay eager (a handful of
    launches).  What it buys: a step issues ~2,000 kernel launches from Python, about 16 ms of host time -- hidden behind 19 ms
    of GPU work at the full single-GPU batch, but the whole story at the strong-scaled per-rank batch of an 8-GPU run
    (1/8 of the rows: 6 ms of GPU work; profiles/r04_graph_probe.txt: 17.8 -> 6.4 ms per step).  What it needs, and how it is
    met: (i) static input buffers (each batch is copied into the captured step's own tensors); (ii) dropout masks that change
    from step to step although a replay repeats its kernel arguments -- a device step counter mixed into every seed
    (``vlpet_set_seed_counter``; torch's own dropout kernels take their Philox offset from the graph-registered generator);
    (iii) no collective inside the capture -- the gradient buckets are reduced by ``FlatGrads.finish`` after the replay
    (``FlatGrads.defer``), at the price of the overlap with the backward (24 MB of fp32 gradients: ~0.2 ms on xGMI);
    (iv) every weight-derived buffer refreshed without the Python forward: the PET fragment packs by ``repack_all(True)`` right
    after the optimizer kernel (into their existing buffers), the epoch-keyed caches (K4 pack, IO-dtype copies of trainable
    weights) by their own rebuild kernels, which are stale at capture time and therefore part of the graph.
    Not captured: per-task adapters / LoRA (the active parameter set is host state) -- such trainers stay eager."""

    def __init__(self, model: nn.Module, config, lr=1e-3, clip=5.0, total_steps=1000, warmup_ratio=0.1,
                 world_size=1, n_buckets=3, process_group=None, overlap_wgrad=False, force_collectives=False, graph=False,
                 capture_collectives=False):
        self.model, self.config, self.clip, self.base_lr = model, config, clip, lr
        on_gpu = next(model.parameters()).is_cuda
        if on_gpu:
            from . import functional as VF
            VF.WGRAD_STREAM = torch.cuda.Stream() if overlap_wgrad else None
        self.flat = FlatGrads(model, world_size, n_buckets, process_group, flatten_params=True, sinks=on_gpu,
                              force_collectives=force_collectives)
        if on_gpu:
            self.optim = FusedAdamW(self.flat, lr=lr, max_norm=clip)
        elif CPU_OPTIMIZER_FACTORY is not None:
            self.optim = CPU_OPTIMIZER_FACTORY(self.flat, lr, clip)
        else:
            raise RuntimeError("vl-pet_amd: the trainer's optimizer step runs on the GPU only")
        self.warmup, self.total = int(total_steps * warmup_ratio), total_steps
        self.step_idx = 0
        self.flat.begin_step(zero=True)
        self.graph = False
        self._graphs, self._graph_seen, self._graph_pool = collections.OrderedDict(), {}, None
        self.max_graphs = MAX_GRAPHS
        self.seed_ctr = None
        # graph mode under data parallelism: False = the three buckets are exchanged by finish() after the replay (serial: ~5 % of a 5.6 ms
        # per-rank step); True = the bucket all-reduces are launched from INSIDE the captured backward, in readiness order, and become
        # nodes of the graph on the collective library's stream (fork at the bucket's last gradient, join before the optimizer) -- the
        # overlap north_star asks for, kept under replay.  Needs a backend whose collectives can be stream-captured (RCCL / "nccl").
        # Tested with a ONE-rank RCCL communicator (tests/test_gpu_dp.py); no multi-GPU box was available to this build, so it is opt-in.
        self.capture_collectives = False
        if capture_collectives and self.flat.dp:
            try:
                self.capture_collectives = dist.get_backend(self.flat.group) == "nccl"
            except Exception:
                self.capture_collectives = False
        self._exchange_in_graph = False
        if graph:
            self.enable_graph()

    # ---- captured steps
    def enable_graph(self):
        if not self.flat.flat.is_cuda:
            raise RuntimeError("vl-pet_amd: Trainer(graph=True) needs the GPU product path")
        if self.flat.per_task or overlap_stream_active():
            return False                 # host-side state decides what a step does: stay eager
        from . import _lib
        if self.seed_ctr is None:
            self.seed_ctr = torch.zeros(1, dtype=torch.int64, device=self.flat.flat.device)
            _lib.check(_lib.load().vlpet_set_seed_counter(self.seed_ctr.data_ptr()), "vlpet_set_seed_counter")
            global _REGISTERED_SEED_CTR
            _REGISTERED_SEED_CTR = self.seed_ctr.data_ptr()
        self.optim.repack_every_pair = True
        self.graph = True
        return True

    def disable_graph(self):
        self.graph = False

    def close(self):
        """Unregister this trainer's device step counter from the library (the pointer is process-wide state: a trainer that goes
        away while registered would leave the dropout kernels reading freed memory -- harmless numerically, wrong in principle)."""
        global _REGISTERED_SEED_CTR
        if self.seed_ctr is not None:
            try:
                if _REGISTERED_SEED_CTR == self.seed_ctr.data_ptr():      # (a later trainer may have registered its own)
                    from . import _lib
                    _lib.load().vlpet_set_seed_counter(None)
                    _REGISTERED_SEED_CTR = None
            except Exception:
                pass
            self.seed_ctr = None
        self.graph = False
        self._graphs.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _leaves(batch):
        out = []
        for k in sorted(batch):
            v = batch[k]
            if torch.is_tensor(v):
                out.append((k, None, v))
            elif isinstance(v, (tuple, list)):
                out += [(k, i, t) for i, t in enumerate(v) if torch.is_tensor(t)]
        return out

    @staticmethod
    def _signature(batch):
        return (batch["task"], bool(batch.get("no_padding", False)),
                tuple((k, i, tuple(t.shape), t.dtype, tuple(t.stride())) for k, i, t in Trainer._leaves(batch)))

    def _fwd_bwd(self, batch) -> torch.Tensor:
        per_token, _ = self.model(batch["input_ids"], batch["vis_inputs"], batch["labels"], batch["task"],
                                  attention_mask=batch.get("attention_mask"), no_padding=bool(batch.get("no_padding", False)))
        loss = task_loss(per_token, batch["labels"], batch.get("scores"), batch["task"])
        if self.flat.flat.is_cuda:
            from . import functional as VF
            # parameter-gradient reductions of this backward (bias column sums, LayerNorm gradients) as one batched launch at its
            # end -- unless bucket all-reduces start from inside the backward (they would read the buckets before the flush)
            VF.DEFER_REDUCES = DEFER_PARAM_REDUCES and ((not self.flat.dp) or self.flat.defer)
            # ... and the column-parallel backward passes sum their row-chunk partials inside the launch (csrc/cols_reduce.h) only while
            # nothing else runs beside the backward: with bucket all-reduces overlapping it (or the weight gradients on a side stream) a
            # workgroup waiting for partners that cannot start would hold its CU for the collective's duration -- the two-launch form then
            from . import _lib
            was = _lib.load().vlpet_set_in_launch_reduce(
                1 if IN_LAUNCH_REDUCE and VF.WGRAD_STREAM is None and ((not self.flat.dp) or self.flat.defer) else 0)
            try:
                loss.backward()
                VF.flush_reduces()
            finally:
                _lib.load().vlpet_set_in_launch_reduce(was)
                VF.DEFER_REDUCES = False
                VF.discard_pending()              # (empty after a clean flush)
                VF._FIN_KEEP.clear()
        else:
            loss.backward()
        return loss

    def _finish_step(self, exchanged: bool = False):
        if not exchanged:                                 # (exchanged: the replayed graph contained the bucket all-reduces and their joins)
            self.flat.finish(average=False)
        self.optim.step(lr_at(self.step_idx, self.base_lr, self.warmup, self.total))   # clips, updates, zeroes the grads
        self.step_idx += 1
        self.flat.begin_step(zero=False)
        if LABEL_CHECK_EVERY and self.step_idx % LABEL_CHECK_EVERY == 0:
            self.check_labels()
        if self.flat.flat.is_cuda and self.step_idx % tail_recheck_every() == 0:
            # K5's backward form per trainable LayerNorm (tail.needs_prenorm), all of them in one reduction + one host read; a captured
            # step has that form baked in, so a flipped decision drops the graphs (the next step of each shape re-captures)
            from . import tail as _tail
            if _tail.recheck_trainable_norms(self.model) and getattr(self, "_graphs", None):
                self._graphs.clear()

    def check_labels(self):
        """Raise if the loss kernels have met labels outside the vocabulary (other than ignore_index -100) since the process started:
        they are treated as ignored tokens, i.e. the run would silently train on fewer tokens (lmloss.bad_label_count; one 4-byte
        device read, every LABEL_CHECK_EVERY steps; the tally is process-wide, so close() does not raise on it -- call this at the end of
        a run)."""
        from . import lmloss, visproj
        tiles = visproj.gemm_exchange_status()
        if tiles:       # (repaired inside the same call: no step ran on wrong rows -- a performance note, not an error)
            import warnings
            warnings.warn(f"vl-pet_amd: {tiles} visual-projection tile(s) gave up waiting for a partner workgroup's LayerNorm statistics and were "
                          "re-normalised by the call's repair pass (csrc/visproj_gemm.hip: the GPU was shared with another long-running "
                          "kernel); visproj.K4_FORM = 'library' avoids the in-launch exchange on a shared device", RuntimeWarning)
        n = lmloss.bad_label_count()
        if n:
            raise IndexError(f"vl-pet_amd: {n} label(s) outside the LM-head vocabulary (and not ignore_index -100) reached the loss -- "
                             "tokenizer / vocabulary mismatch")

    def _capture(self, key, batch):
        from . import functional as VF
        # Every weight-derived cache must MISS inside the capture, so that its rebuild becomes a node of the graph (a replay runs no
        # Python forward).  They are stale right after an optimizer step -- but not if any forward ran since (validation between two
        # train steps, a no_grad probe): the graph would then bake in buffers nothing refreshes.  A new weights epoch makes them
        # stale by construction; the packs and the IO-dtype shadow, which the optimizer step refreshes itself, are brought to it here.
        VF.bump_weights_epoch()
        VF.repack_all(True)
        if self.flat.flat_p_io is not None:
            self.flat.refresh_io_shadow(self.flat.flat_p_io.dtype)
        while len(self._graphs) >= max(1, int(self.max_graphs)):       # least recently replayed shape goes (its pool blocks are reused)
            self._graphs.popitem(last=False)
        static = {}
        for k, v in batch.items():
            if torch.is_tensor(v):
                static[k] = v.clone()
            elif isinstance(v, (tuple, list)):
                static[k] = type(v)(t.clone() if torch.is_tensor(t) else t for t in v)
            else:
                static[k] = v
        timer, VF.TIMER = VF.TIMER, None            # (event brackets are host-timed launches: not inside a capture)
        self.flat.defer = not self.capture_collectives
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            # thread_local: a collective library's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
                loss = self._fwd_bwd(static).detach()     # (detached: keeping the graph's root would keep its AccumulateGrad nodes, and
                                                          #  their capture stream, alive into later eager steps)
                if self.capture_collectives:              # the buckets left from inside the backward; their joins belong to the graph too
                    self.flat.finish(average=False)
        except Exception as e:      # a model whose step cannot be captured (host sync, data-dependent shape): stay eager, loudly
            import warnings
            warnings.warn(f"vl-pet_amd: capturing the train step failed ({type(e).__name__}: {e}); this trainer continues with eager launches")
            self.graph = False
            self.optim.repack_every_pair = False
            torch.cuda.synchronize()
            self.flat.flat.zero_()
            return None
        finally:
            self.flat.defer = False
            VF.TIMER = timer
            self.flat.begin_step(zero=False)        # (the capture ran the sinks' host-side bookkeeping, not their kernels)
        self._graph_pool = g.pool()
        ent = (g, static, loss)
        self._graphs[key] = ent
        return ent

    def input_buffers(self, batch):
        """The input tensors the captured step of this batch's shape reads -- a dict shaped like ``batch`` -- or None while that shape has
        not been captured (graph mode off, first two steps of a shape).  A replay reads its inputs from these fixed addresses; step()
        copies each batch into them device-to-device (200 MB of CLIP features at configs[1]'s VQA batch: ~70 us).  A loader that lets its
        host-to-device copies land IN these tensors and hands them to step() saves that copy: step() skips every leaf that already is
        the buffer."""
        if not self.graph:
            return None
        ent = self._graphs.get(self._signature(batch))
        return None if ent is None else ent[1]

    def _graph_step(self, batch) -> torch.Tensor:
        key = self._signature(batch)
        ent = self._graphs.get(key)
        self.seed_ctr += 1
        if ent is None:
            seen = self._graph_seen.get(key, 0)
            if len(self._graph_seen) > 64 * max(1, int(self.max_graphs)):      # (ragged data: do not remember every shape ever met)
                self._graph_seen.clear()
            self._graph_seen[key] = seen + 1
            if seen < 1:       # the first step of a shape runs eagerly: lazy initialisation, GEMM solution lookups, the label check
                return self._eager_step_in_graph_mode(batch)
            ent = self._capture(key, batch)
            if ent is None:
                return self._eager_step_in_graph_mode(batch)
        else:
            self._graphs.move_to_end(key)
        g, static, loss = ent
        for (k, i, src), (_, _, dst) in zip(self._leaves(batch), self._leaves(static)):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        g.replay()
        self._finish_step(exchanged=self.capture_collectives)
        return loss.detach().clone()

    def _eager_step_in_graph_mode(self, batch) -> torch.Tensor:
        """An eager step of a trainer that otherwise replays graphs (first sight of a shape, a failed capture).  Under data
        parallelism its gradient buckets are exchanged by finish() in index order, exactly as after a replay: ranks decide eager vs
        replay from their OWN batch shapes, and a rank launching its buckets from inside the backward (readiness order, e.g. 1, 0, 2)
        next to one replaying (0, 1, 2) would issue mismatched collectives."""
        self.flat.defer = not self.capture_collectives      # (captured collectives run in readiness order: so does this step)
        try:
            loss = self._fwd_bwd(batch)
        finally:
            self.flat.defer = False
        self._finish_step()
        return loss.detach()

    def step(self, batch) -> torch.Tensor:
        if self.graph:
            return self._graph_step(batch)
        loss = self._fwd_bwd(batch)
        self._finish_step()
        return loss.detach()


def overlap_stream_active() -> bool:
    """Weight gradients on a side stream (Trainer(overlap_wgrad=True)) are not captured."""
    from . import functional as VF
    return VF.WGRAD_STREAM is not None
