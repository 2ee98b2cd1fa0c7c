"""torch.autograd glue over the C ABI (device memory, streams and autograd only -- the
arithmetic lives in csrc/).  Every function here requires CUDA(ROCm) tensors and raises if the
HIP library is unavailable; there is deliberately no eager fallback."""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib

GATE_NONE, GATE_MUL, GATE_ADD = _lib.GATE_NONE, _lib.GATE_MUL, _lib.GATE_ADD


class KernelTimer:
    """Optional HIP-event bracketing of individual launches (bench.py's live roofline measurement).
    Events are recorded on the stream the kernels are enqueued on (torch's current stream)."""

    def __init__(self, names=None):
        self.records = []      # (name, rows, start_event, end_event)
        self.names = None if names is None else frozenset(names)     # bracket only these launch groups (None: all)

    def wants(self, name):
        return self.names is None or name in self.names

    def bracket(self, name, rows, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.records.append((name, rows, e0, e1))
        return out

    def summary(self):
        """name -> dict(launches, rows, total_us, by_rows = {rows of a launch: [launches, total_us]}); call after torch.cuda.synchronize()."""
        agg = {}
        for name, rows, e0, e1 in self.records:
            a = agg.setdefault(name, dict(launches=0, rows=0, total_us=0.0, by_rows={}))
            us = e0.elapsed_time(e1) * 1e3
            a["launches"] += 1
            a["rows"] += rows
            a["total_us"] += us
            b = a["by_rows"].setdefault(int(rows), [0, 0.0])
            b[0] += 1
            b[1] += us
        return agg


TIMER: Optional[KernelTimer] = None

# Optional side stream for the weight-gradient half of the K1 backward (train.Trainer(overlap_wgrad=True); OFF by
# default: on one MI355X the step is GPU-bound and the overlap measured 5 % slower, 15.2 k vs 16.0 k samples/s).
# The row-parallel half stays
# on the autograd stream -- the next backward op needs dx1 / dx2 -- while the column-parallel weight gradients, which
# nobody reads before the optimizer step, run concurrently with the frozen backward that follows.  Only used when
# every weight gradient of the call goes straight into the trainer's flat buffer (train.GradSink).
WGRAD_STREAM: Optional["torch.cuda.Stream"] = None


# Deferred reductions of parameter gradients (bias column sums, LayerNorm dgamma / dbeta): a trainer sets DEFER_REDUCES for its
# backward and calls flush_reduces() once after it -- one batched launch (vlpet_reduce_batch) instead of one 5.6-us launch per
# parameter (127 per step in the LoRA runs, which train every bias).  Only gradients written straight into the trainer's flat buffer
# are deferred (train.GradSink: nobody reads them before the optimizer); with bucketed all-reduces running from inside the backward the
# trainer leaves the switch off.
DEFER_REDUCES = False
_PENDING_REDUCES: list = []
# ... and, round 5, the finalize launches of the weight-gradient kernels (the sum of a call's row-chunk partials into dW / db): with
# the switch on, a backward op whose gradients all go to sinks queues that launch inside the library (vlpet_finalize_defer) and
# flush_reduces() issues the queue, 16 calls per launch (19 launches -> 2 in a BART-base step, 37 -> 3 in T5-base).  The call's
# workspace (the partial sums) is kept alive here until then.  Bit-identical (tests/test_gpu_graph.py) -- and measured worth NOTHING:
# BART-base 17.07 / 17.65 vs 17.19 / 16.96 ms per step, emulated rank 1 of 8 5.34 vs 5.35 ms, T5-base 24.2 vs 24.2, 9.30 vs 9.27
# (profiles/r05_deferred_finalize_ab.txt): the batched launch reads the same partial sums, by then out of the caches.  Off by default.
DEFER_FINALIZE = False
_PENDING_FINALIZE_KEEP: list = []


class _deferred_finalize:
    """``with _deferred_finalize(all_sinks, ws, ...):`` around a backward call -- its finalize launch is queued when a trainer defers."""

    __slots__ = ("on", "keep")

    def __init__(self, all_sinks: bool, *keep):
        self.on = bool(DEFER_REDUCES and DEFER_FINALIZE and all_sinks)
        self.keep = keep

    def __enter__(self):
        if self.on:
            _lib.load().vlpet_finalize_defer(1)
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib.load().vlpet_finalize_defer(0)
            _PENDING_FINALIZE_KEEP.extend(self.keep)
        return False


# Under graph capture the finalize launch of a gated K1 backward (12 / 24 per step; ~10 us of a few dozen workgroups each) is issued on a
# side stream: a parallel branch of the captured graph, joined by flush_reduces() at the end of the backward.  Nobody reads a weight
# gradient before that, and the partial sums it reads are still warm (unlike the deferred + batched form above, which read them cold).
# Eager steps keep it in line: two stream switches per op cost the host more than the launch hides.
# MEASURED (profiles/r05_finalize_side_stream_ab.txt): the replayed step gets SLOWER -- 6.02 vs 5.37 ms (BART, emulated rank 1 of 8), 10.45 vs 9.17 ms
# (T5), 24.9 vs 24.0 ms at T5's full batch: a fork + join inside a hipGraph costs ~50 us here (12 / 24 of them per step), five times the
# launch it hides.  Off.
FINALIZE_SIDE_STREAM = False
_FIN_STREAM = None
_FIN_PENDING = False
_FIN_KEEP: list = []


def _finalize_on_side(fn, *keep):
    """``fn(stream_handle)`` = the finalize-only call of a backward op, issued on the side stream after everything queued on the current one."""
    global _FIN_STREAM, _FIN_PENDING
    cur = torch.cuda.current_stream()
    if _FIN_STREAM is None:
        _FIN_STREAM = torch.cuda.Stream()
    _FIN_STREAM.wait_stream(cur)
    with torch.cuda.stream(_FIN_STREAM):
        rc = fn(_FIN_STREAM.cuda_stream)
    _FIN_PENDING = True
    _FIN_KEEP.extend(keep)            # (the partial sums live in the op's workspace: not to be recycled before the join)
    return rc


def join_finalize_stream():
    global _FIN_PENDING
    if _FIN_PENDING:
        torch.cuda.current_stream().wait_stream(_FIN_STREAM)
        _FIN_PENDING = False
    _FIN_KEEP.clear()


def discard_pending():
    """Error path of a trainer's backward: nothing queued survives the step."""
    _PENDING_REDUCES.clear()
    if _PENDING_FINALIZE_KEEP:
        _PENDING_FINALIZE_KEEP.clear()
    lib = _lib.load()
    if lib.vlpet_finalize_pending():
        lib.vlpet_finalize_discard()


def reduce_partials(part: torch.Tensor, nb: int, d: int, out0: Optional[torch.Tensor], out1: Optional[torch.Tensor], deferrable: bool):
    """``part`` viewed as [nb][2 d] -> out0 [d], out1 [d] (fp32, overwritten; either may be None): now, or at flush_reduces()."""
    if DEFER_REDUCES and deferrable:
        _PENDING_REDUCES.append((part, int(nb), int(d), out0, out1))
        return
    rc = _lib.load().vlpet_sublayer_tail_reduce(part.data_ptr(), int(nb), int(d), _ptr(out0), _ptr(out1), _stream())
    _lib.check(rc, "vlpet_sublayer_tail_reduce")


def flush_reduces():
    lib = _lib.load()
    join_finalize_stream()
    if lib.vlpet_finalize_pending():
        rc = lib.vlpet_finalize_flush(_stream())
        _PENDING_FINALIZE_KEEP.clear()
        _lib.check(rc, "vlpet_finalize_flush")
    if not _PENDING_REDUCES:
        return 0
    jobs, n = list(_PENDING_REDUCES), len(_PENDING_REDUCES)
    _PENDING_REDUCES.clear()
    vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
    rc = _lib.load().vlpet_reduce_batch(vp(*[j[0].data_ptr() for j in jobs]), vp(*[_ptr(j[3]) for j in jobs]), vp(*[_ptr(j[4]) for j in jobs]),
                                        ip(*[j[1] for j in jobs]), ip(*[j[2] for j in jobs]), n, _stream())
    _lib.check(rc, "vlpet_reduce_batch")
    return n


def _timed(name, rows, fn):
    if TIMER is None or not TIMER.wants(name):
        return fn()
    return TIMER.bracket(name, rows, fn)


def _io_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.VLPET_BF16
    if t.dtype == torch.float32:
        return _lib.VLPET_F32
    raise RuntimeError(f"vl-pet_amd: unsupported activation dtype {t.dtype} (bf16 or fp32)")


def _param_dtype(t: torch.Tensor) -> int:
    return _io_dtype(t)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vl-pet_amd: the PET hot path runs on the GPU only (got a CPU tensor)")


def rank_tiles(r: int) -> int:
    t = _lib.load().vlpet_rank_tiles(int(r))
    if t < 0:
        _lib.check(t, f"rank_tiles({r})")
    return t


class PackedPair:
    """MFMA-fragment pack of one (down, up) projection pair (see tests/packing_spec.py / csrc/pack.hip)."""

    __slots__ = ("buf", "tiles", "r", "d", "io_dtype")

    def __init__(self, buf, tiles, r, d, io_dtype):
        self.buf, self.tiles, self.r, self.d, self.io_dtype = buf, tiles, r, d, io_dtype


def pack_pair(down_w: Sequence[torch.Tensor], down_b: Optional[Sequence[torch.Tensor]],
              up_w: torch.Tensor, up_b: Optional[torch.Tensor], io_dtype: int,
              tiles: Optional[int] = None, out: Optional["PackedPair"] = None) -> PackedPair:
    """``out``: an earlier pack of the same geometry to refresh IN PLACE (its buffer may be baked into a captured graph: a cache that
    answered a miss with a new buffer would leave the graph reading the old one -- or freed memory)."""
    lib = _lib.load()
    _need_cuda(up_w, *down_w)
    n = len(down_w)
    rh, d = down_w[0].shape
    r = rh * n
    if tuple(up_w.shape) != (d, r):
        raise RuntimeError(f"vl-pet_amd: up weight {tuple(up_w.shape)} does not match down {r}x{d}")
    if tiles is None:
        tiles = rank_tiles(r)
    ws = [w.detach().contiguous() for w in down_w]
    bs = [b.detach().contiguous() for b in down_b] if down_b is not None else None
    uw = up_w.detach().contiguous()
    ub = up_b.detach().contiguous() if up_b is not None else None
    pd = _param_dtype(uw)
    for t in ws + (bs or []) + ([ub] if ub is not None else []):
        if _param_dtype(t) != pd:
            raise RuntimeError("vl-pet_amd: mixed parameter dtypes in one projection pair")
    nbytes = lib.vlpet_packed_bytes(tiles, d, io_dtype)
    if (out is not None and out.buf.numel() == nbytes and out.buf.device == uw.device and out.tiles == tiles and out.r == r
            and out.d == d and out.io_dtype == io_dtype):
        buf = out.buf
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=uw.device)
    arr_w = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    arr_b = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs]) if bs is not None else None
    rc = lib.vlpet_pack_pair(arr_w, arr_b, n, uw.data_ptr(), _ptr(ub), r, d, tiles, pd, io_dtype,
                             buf.data_ptr(), _stream())
    _lib.check(rc, "vlpet_pack_pair")
    return PackedPair(buf, tiles, r, d, io_dtype)


# Bumped by optimizers that update parameters behind autograd's back (the fused flat-buffer AdamW writes through
# raw pointers, which does not touch ``Tensor._version``); part of every pack-cache key.
# keep the gated forward's bottleneck activations for the backward (vlpet_adapter_gate_fwd_save / _bwd_saved); False = the
# backward recomputes them from x1, x2 (no extra memory between forward and backward)
SAVE_ACTIVATIONS = True
# A/B switch for benches and tests: True = the round-2 split of the gated K1 backward (row kernel that also writes dh / dq +
# streaming weight-gradient kernel, ABI phases bit 2) instead of pass 1 + the column-parallel pass
K1_BWD_PREVIOUS_SPLIT = False
# round 5: the multiplicative gate's backward starts from the forward's OUTPUT y = gs * h * g when the forward saved its
# activations: dq = dy * y * (1 - g), so pass 1 recomputes the gate's up projection only (vlpet_adapter_gate_bwd_saved_y).
# False = from x2 (recompute h = s2 * x2 + sd * up_A(z_a)), the rounds 2-4 form.
K1_BWD_FROM_OUTPUT = True
# round 6: pass 2 of the gated backward (r <= 96) sums its row-chunk partials inside the launch (csrc/cols_reduce.h).  True = the
# round-3 form (partial slabs + a finalize launch, ABI phases bit 5) for same-box A/Bs; the results are bit-identical.  (The
# library-wide switch vlpet_set_in_launch_reduce -- train.Trainer turns it off where gradient collectives overlap the backward --
# covers K2 / K3 as well.)
K1_BWD_FINALIZE_LAUNCH = False

WEIGHTS_EPOCH = 0
# Part of the keys of the derived copies of FROZEN tensors (fused q|k|v weight, padded LM head, fp32 LayerNorm copies, the
# logits-bias flag).  Those keys also carry (data_ptr, Tensor._version), but ``p.data`` has its own version counter: in-place
# edits through ``.data`` (``m.weight.data.normal_()``, ``p.data.copy_()`` in hand-written checkpoint loaders) or through raw
# pointers are invisible to them.  ``invalidate_caches()`` is the explicit way out; the hosts call it from a
# ``load_state_dict`` post-hook.  (Not bumped by optimizer steps: frozen tensors do not change there.)
FROZEN_EPOCH = 0


def bump_weights_epoch():
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


def invalidate_caches():
    """Drop every derived copy of a parameter (weight packs and the caches of frozen tensors): call after changing weights in
    a way autograd's version counters do not see (``.data`` edits, raw-pointer writes, custom checkpoint loaders)."""
    global FROZEN_EPOCH
    FROZEN_EPOCH += 1
    bump_weights_epoch()


_PACK_CACHES = None       # weak set of the PackCache objects that packed parameters directly (see repack_all)


class PackCache:
    """Re-pack only when a parameter changed (optimizer steps bump ``Tensor._version`` or ``WEIGHTS_EPOCH``)."""

    def __init__(self):
        self._key = None
        self._val = None
        self._src = None      # (down_w, down_b, up_w, up_b, io_dtype, tiles, epoch of last use): what repack_all re-packs

    @staticmethod
    def _make_key(down_w, down_b, up_w, up_b, io_dtype, tiles):
        ts = list(down_w) + (list(down_b) if down_b is not None else []) + [up_w] + ([up_b] if up_b is not None else [])
        return (io_dtype, tiles, WEIGHTS_EPOCH) + tuple((t.data_ptr(), t._version) for t in ts)

    def get(self, down_w, down_b, up_w, up_b, io_dtype, tiles=None) -> PackedPair:
        key = self._make_key(down_w, down_b, up_w, up_b, io_dtype, tiles)
        if key != self._key:
            self._val = pack_pair(down_w, down_b, up_w, up_b, io_dtype, tiles, out=self._val)
            self._key = key
        global _PACK_CACHES
        if _PACK_CACHES is None:
            import weakref
            _PACK_CACHES = weakref.WeakSet()
        _PACK_CACHES.add(self)
        self._src = (list(down_w), list(down_b) if down_b is not None else None, up_w, up_b, io_dtype, tiles, WEIGHTS_EPOCH)
        return self._val

    def get_derived(self, sources, tag, build, io_dtype, tiles=None) -> PackedPair:
        """Pack of tensors DERIVED from parameters (slices, concatenations, ``.contiguous()`` copies): the key is made from
        the source parameters (+ ``tag``, the derivation's geometry), never from the temporaries -- their ``_version`` is
        always 0 and their address is whatever the caching allocator hands out -- and ``build()`` -> (down_w, down_b,
        up_w, up_b) runs only on a miss."""
        key = (io_dtype, tiles, WEIGHTS_EPOCH, tag) + tuple((t.data_ptr(), t._version) for t in sources)
        if key != self._key:
            self._val = pack_pair(*build(), io_dtype, tiles, out=self._val)
            self._key = key
        return self._val


def repack_all(every_pair: bool = False) -> int:
    """Re-pack, in a few batched launches (vlpet_pack_pairs: 8 pairs each), every projection pair that was used since the
    previous call -- what a trainer does right after its optimizer step, instead of ~30 single 7.6-us pack launches spread over
    the next forward.  Pairs are re-packed INTO their existing buffers and their cache keys moved to the current weights
    epoch, so the next ``PackCache.get`` is a hit; anything that does not fit (contiguity, mixed dtypes, a parameter that
    was replaced) is simply left to the lazy path.  Returns the number of pairs packed."""
    if not _PACK_CACHES:
        return 0
    lib = _lib.load()
    groups = {}
    for c in list(_PACK_CACHES):
        src, val = c._src, c._val
        if src is None or val is None:
            continue
        down_w, down_b, up_w, up_b, io_dtype, tiles, used = src
        if used < WEIGHTS_EPOCH - 1 and not every_pair:   # not used in the step that just ended: leave it to the lazy path
            continue                                      # (every_pair: a trainer replaying captured steps runs no Python forward)
        ts = down_w + (down_b or []) + [up_w] + ([up_b] if up_b is not None else [])
        if not all(t.is_cuda and t.is_contiguous() for t in ts) or len({t.dtype for t in ts}) != 1:
            continue
        n, (rh, d) = len(down_w), down_w[0].shape
        if tuple(up_w.shape) != (d, rh * n) or val.d != d or val.tiles != (tiles if tiles is not None else val.tiles):
            continue
        g = (n, rh * n, d, val.tiles, _param_dtype(up_w), io_dtype, down_b is not None, up_b is not None, up_w.device)
        groups.setdefault(g, []).append(c)
    done = 0
    for (n, r, d, tiles, pd, io_dtype, has_bd, has_bu, dev), caches in groups.items():
        k = len(caches)
        wd = (ctypes.c_void_p * (k * n))(*[w.data_ptr() for c in caches for w in c._src[0]])
        bd = (ctypes.c_void_p * (k * n))(*[b.data_ptr() for c in caches for b in c._src[1]]) if has_bd else None
        wu = (ctypes.c_void_p * k)(*[c._src[2].data_ptr() for c in caches])
        bu = (ctypes.c_void_p * k)(*[c._src[3].data_ptr() for c in caches]) if has_bu else None
        out = (ctypes.c_void_p * k)(*[c._val.buf.data_ptr() for c in caches])
        with torch.cuda.device(dev):
            rc = lib.vlpet_pack_pairs(k, wd, bd, n, wu, bu, r, d, tiles, pd, io_dtype, out, _stream())
        _lib.check(rc, "vlpet_pack_pairs")
        for c in caches:
            down_w, down_b, up_w, up_b, io, tl, _ = c._src
            c._key = PackCache._make_key(down_w, down_b, up_w, up_b, io, tl)
        done += k
    return done


def _flat(x: torch.Tensor, d: int) -> torch.Tensor:
    x = x.reshape(-1, d)
    return x if x.is_contiguous() else x.contiguous()


def _grad_like(g32: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    return g32 if p.dtype == torch.float32 else g32.to(p.dtype)


def _empty_result(x: torch.Tensor, params) -> torch.Tensor:
    """Zero-row input: nothing to launch.  The (empty) result stays connected to every parameter so that
    their gradients are zeros rather than None, like the reference's op chain on an empty batch."""
    out = x.clone()
    for p in params:
        if p is not None and p.requires_grad:
            out = out + (p.reshape(-1)[:1].sum() * 0).to(out.dtype)
    return out


def _grad_dest(p: torch.Tensor, shape, block: bool = False):
    """Where a weight-gradient kernel writes: the parameter's slot of the trainer's flat gradient buffer when it
    offers one for this step (train.GradSink: first write of the step, fp32, matching shape) -- then autograd
    gets ``None`` for that input and no accumulate kernel runs -- else a fresh fp32 tensor returned to autograd."""
    sink = getattr(p, "_vlpet_block_sink" if block else "_vlpet_sink", None)
    if sink is not None and tuple(sink.view.shape) == tuple(shape) and sink.view.device == p.device:
        v = sink.take()
        if v is not None:
            return v, sink
    return torch.empty(*shape, dtype=torch.float32, device=p.device), None


def _finish(dests):
    """Autograd return values for (tensor, sink, param) triples: None where the kernel wrote into a sink."""
    out = []
    for t, sink, p in dests:
        if sink is not None:
            sink.done()
            out.append(None)
        else:
            out.append(_grad_like(t, p))
    return out


FANOUT_SUM = True          # A/B switch (tools/ab_switches.py: VLPET_NO_FANOUT_SUM=1): fanout() below


class _FanoutFn(torch.autograd.Function):
    """``n`` aliases of one tensor for ``n`` consumers, so that their gradients arrive TOGETHER and are summed in one launch
    (vlpet_sum_n: n + 1 row units, fp32 accumulation, one rounding) instead of autograd's pairwise accumulation (n - 1 passes of three
    units, a bf16 rounding each).  The encoder output under the decoder layers' cross-attention is the case
    (my_transformers/modeling_bart.py:2300-2330: every layer takes encoder_hidden_states)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        g0 = live[0]
        same = all(g.dtype == g0.dtype and g.shape == g0.shape and g.is_contiguous() for g in live)
        if not (same and g0.is_cuda and g0.dtype in (torch.bfloat16, torch.float32) and g0.numel() % 8 == 0 and g0.numel() > 0):
            total = live[0]
            for g in live[1:]:
                total = total + g
            return total, None
        import ctypes
        lib = _lib.load()
        out = torch.empty_like(g0)
        arr = (ctypes.c_void_p * len(live))(*[g.data_ptr() for g in live])
        rc = _timed("fanout_sum", g0.numel() // g0.shape[-1], lambda: lib.vlpet_sum_n(arr, len(live), out.data_ptr(), g0.numel(), _io_dtype(g0), _stream()))
        _lib.check(rc, "vlpet_sum_n")
        return out, None


def fanout(x: torch.Tensor, n: int):
    """``n`` references to ``x`` for ``n`` consumers; on the GPU with gradients on, aliases whose gradients are summed in one launch."""
    if n <= 1 or not FANOUT_SUM or not x.is_cuda or not (x.requires_grad and torch.is_grad_enabled()):
        return (x,) * max(1, n)
    return _FanoutFn.apply(x, n)


class _CrossKeyProjFn(torch.autograd.Function):
    """``n`` frozen projections of ONE input as ONE GEMM: ``x [.., E] -> [.., n * E]``, handed out as its n column blocks (views).  The
    decoder layers' cross-attention key projections of the encoder output (my_transformers/modeling_bart.py:2300-2330 hands every layer
    the same encoder_hidden_states; each layer's ``encoder_attn.k_proj`` projects it, :425) -- n small GEMMs that all read x, and in
    the backward n dgrad GEMMs that all accumulate into d/dx.  Here: one GEMM forward; backward, the attention calls write their dk
    into the blocks of one gradient buffer (attention.KeyGradSlot) and ONE GEMM with K = n * E produces d/dx."""

    @staticmethod
    def forward(ctx, x, w_all, b_all, n, slot):
        y = F.linear(x, w_all, b_all)
        E = w_all.shape[0] // n
        ctx.save_for_backward(w_all)
        ctx.slot, ctx.n, ctx.xshape = slot, n, x.shape
        return tuple(y[..., i * E:(i + 1) * E] for i in range(n))

    @staticmethod
    def backward(ctx, *dks):
        (w_all,) = ctx.saved_tensors
        n, slot = ctx.n, ctx.slot
        ctx.slot = None
        E = w_all.shape[0] // n
        buf = slot.buf if slot is not None else None
        if slot is not None:
            slot.buf = None
        whole = buf is not None and all(
            dk is not None and dk.data_ptr() == buf.data_ptr() + i * E * buf.element_size() and dk.shape[:-1] == buf.shape[:-1]
            and dk.stride() == buf.stride() for i, dk in enumerate(dks))
        if whole:
            dx = buf.view(-1, n * E) @ w_all
        else:                       # (a consumer that went another way: per-block products)
            dx = None
            for i, dk in enumerate(dks):
                if dk is None:
                    continue
                t = dk.reshape(-1, E).to(w_all.dtype) @ w_all[i * E:(i + 1) * E]
                dx = t if dx is None else dx.add_(t)
        return (None if dx is None else dx.view(ctx.xshape)), None, None, None, None


def cross_key_blocks(x: torch.Tensor, w_all: torch.Tensor, b_all: torch.Tensor, n: int):
    """(k_0 .. k_{n-1}, slot): the n key projections of ``x`` as column blocks of one GEMM's output and the KeyGradSlot their attention
    calls pass on (``short_attention(..., k_slot=(slot, i))``)."""
    from .attention import KeyGradSlot
    slot = KeyGradSlot(n, w_all.shape[0] // n) if (x.requires_grad and torch.is_grad_enabled()) else None
    return _CrossKeyProjFn.apply(x, w_all, b_all, n, slot), slot


class ResidualLink:
    """Hand-over of the residual-stream gradient between the two ops that read the sublayer input x1: the gate of K1 and
    the sublayer tail LayerNorm(x1 + dropout(y)) (my_transformers/modeling_bart.py:1196, 1259-1261).  Autograd would sum
    their two contributions to d/dx1 with a separate elementwise pass over [M, d]; with a link the tail's backward (which
    runs first: its output is downstream of K1's) parks its dx1 here and returns no gradient for x1, and K1's backward
    adds it inside its kernel (vlpet_adapter_gate_bwd_saved_acc) and returns the sum.  Armed by K1's forward only when its
    backward will take the hand-over (gated, saved-activation form)."""

    __slots__ = ("armed", "dx1", "shared")

    def __init__(self):
        self.armed = False
        self.dx1 = None
        self.shared = False        # the parked tensor is also somebody else's gradient: a consumer must not write into it


class _LinearAccFn(torch.autograd.Function):
    """``n`` frozen projections of ONE input (``x -> x W_i^T + b_i``) whose input gradient starts from a parked one.

    The input of a sublayer's first GEMM is read by a second op further down (the gate of K1, the sublayer tail's
    residual add, the value-parallel adapter of K2: my_transformers/modeling_bart.py:1147-1155, 1259-1261, 427-430), and
    autograd sums the two input gradients with an elementwise pass over ``[M, d]``.  That other op is downstream of this
    GEMM's output, so its backward always runs first: it parks its gradient in the ``ResidualLink`` this forward armed and
    returns none, and the dgrad GEMM here accumulates onto it (``C += dY W``, beta = 1 in the library GEMM's epilogue) --
    one gradient for x, no add kernel.  Weights and biases are frozen (no gradient is produced for them)."""

    @staticmethod
    def forward(ctx, x, link, *wb):
        n = len(wb) // 2
        ws, bs = wb[:n], wb[n:]
        ctx.link = None
        if link is not None and ctx.needs_input_grad[0]:
            link.armed = True
            ctx.link = link
        ctx.save_for_backward(*ws)
        ctx.xshape = x.shape
        outs = tuple(F.linear(x, w, b) for w, b in zip(ws, bs))
        return outs if n > 1 else outs[0]

    @staticmethod
    def backward(ctx, *dys):
        ws = ctx.saved_tensors
        link, ctx.link = ctx.link, None
        base = None
        shared = False
        if link is not None:
            base, link.dx1, shared = link.dx1, None, link.shared
            link.shared = False
        K = ctx.xshape[-1]
        acc = None
        if base is not None:
            if base.dtype == ws[0].dtype and base.is_contiguous() and base.numel() == math.prod(ctx.xshape):
                acc = base.view(-1, K)
                if shared:
                    acc = acc.clone()
            else:                       # (never on the product path: kept correct rather than fast)
                acc = base.reshape(-1, K).to(ws[0].dtype).contiguous()
        for dy, w in zip(dys, ws):
            if dy is None:
                continue
            dy2 = dy.reshape(-1, dy.shape[-1])
            if dy2.dtype != w.dtype:
                dy2 = dy2.to(w.dtype)
            if acc is None:
                acc = dy2 @ w
            else:
                acc.addmm_(dy2, w)
        dx = acc.view(ctx.xshape) if acc is not None else None
        return (dx, None) + (None,) * (2 * len(ws))


def io_view(p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``p`` in the activation dtype: the parameter's slice of its trainer's IO-dtype shadow buffer when that is current
    (train.FlatGrads.refresh_io_shadow: one cast launch per optimizer step for all trainable parameters), else a cast."""
    if p.dtype == dtype:
        return p
    sh = getattr(p, "_vlpet_io", None)
    if sh is not None and sh[0].io_epoch == WEIGHTS_EPOCH and sh[1].dtype == dtype:
        return sh[1]
    return p.to(dtype)


class _LinearTrainBiasFn(torch.autograd.Function):
    """``x W^T + b`` with a FROZEN weight and a TRAINABLE bias (the reference's LoRA runs unfreeze every bias next to the frozen
    projections): the bias gradient = column sums of dy through vlpet_colsum (two HIP launches, fp32, straight into the
    parameter's slot of the flat gradient buffer when the trainer offers one) instead of autograd's bf16 ``sum(0)`` + cast +
    accumulate."""

    @staticmethod
    def forward(ctx, x, w, b, link=None):
        ctx.link = None
        if link is not None and ctx.needs_input_grad[0]:
            link.armed = True           # K3's delta on this projection (same x, downstream of this output) parks its d/dx
            ctx.link = link
        ctx.save_for_backward(w, b)
        return F.linear(x, w, io_view(b, x.dtype))

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        w, b = ctx.saved_tensors
        n = dy.shape[-1]
        dy2 = _flat(dy, n)
        if dy2.dtype != w.dtype:
            dy2 = dy2.to(w.dtype)
        dx = None
        link, ctx.link = ctx.link, None
        if ctx.needs_input_grad[0]:
            base = None
            if link is not None:
                base, link.dx1, shared = link.dx1, None, link.shared
                link.shared = False
                if base is not None and (shared or base.dtype != w.dtype or not base.is_contiguous()):
                    base = base.to(w.dtype).contiguous().clone() if shared else base.to(w.dtype).contiguous()
            if base is not None:
                dx = base.view(-1, w.shape[1]).addmm_(dy2, w).view(*dy.shape[:-1], w.shape[1])      # C += dY W in the GEMM epilogue
            else:
                dx = (dy2 @ w).view(*dy.shape[:-1], w.shape[1])
        db = None
        if ctx.needs_input_grad[2]:
            M = dy2.shape[0]
            (t, sink) = _grad_dest(b, (n,))
            nb = lib.vlpet_sublayer_tail_partials(M)
            ws = torch.empty(nb * n, dtype=torch.float32, device=dy2.device)
            if DEFER_REDUCES and sink is not None:       # pass 1 now, the reduction with the step's other ones (flush_reduces)
                rc = lib.vlpet_colsum_partial(dy2.data_ptr(), M, n, ws.data_ptr(), _io_dtype(dy2), _stream())
                _lib.check(rc, "vlpet_colsum_partial")
                tf = t.reshape(-1)
                reduce_partials(ws, nb, n // 2, tf[:n // 2], tf[n // 2:], True)
            else:
                rc = lib.vlpet_colsum(dy2.data_ptr(), M, n, ws.data_ptr(), t.data_ptr(), _io_dtype(dy2), _stream())
                _lib.check(rc, "vlpet_colsum")
            db = _finish([(t, sink, b)])[0]
        return dx, None, db, None


def linear_train_bias_ok(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> bool:
    """Whether ``linear_train_bias`` applies: CUDA bf16 / fp32 activations, frozen weight in that dtype, trainable bias."""
    return (b is not None and b.requires_grad and not w.requires_grad and x.is_cuda and w.dtype == x.dtype
            and x.dtype in (torch.bfloat16, torch.float32) and w.shape[0] % 16 == 0 and w.shape[0] <= (4096 if x.dtype == torch.bfloat16 else 2048)
            and x.numel() > 0 and torch.is_grad_enabled())


def linear_train_bias(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, link: Optional["ResidualLink"] = None) -> torch.Tensor:
    """``link``: shared with the K3 delta applied on top of this projection (lora_delta(..., link=link))."""
    return _LinearTrainBiasFn.apply(x, w, b, link)


def linear_acc(x: torch.Tensor, link: Optional[ResidualLink], *mods):
    """``F.linear`` of ``x`` through each of ``mods`` (``nn.Linear``-like, or ``(weight, bias)`` pairs), frozen, with the
    input gradient accumulated onto whatever the op sharing ``link`` parks (see _LinearAccFn).  One result per module."""
    ws, bs = [], []
    for m in mods:
        w, b = (m.weight, m.bias) if hasattr(m, "weight") else m
        if w.requires_grad or (b is not None and b.requires_grad):
            raise RuntimeError("vl-pet_amd: linear_acc is for frozen projections")
        ws.append(w if w.dtype == x.dtype else w.to(x.dtype))
        bs.append(b if b is None or b.dtype == x.dtype else b.to(x.dtype))
    return _LinearAccFn.apply(x, link, *ws, *bs)


class _AdapterGateFn(torch.autograd.Function):
    """K1.  inputs: x1, x2, then N_h down weights, N_h down biases, up w, up b, gate down w/b, gate up w/b."""

    @staticmethod
    def forward(ctx, x1, x2, pk_a, pk_g, n_heads, gate_mode, delta_scale, x2_scale, gate_scale, link, out_link, *params):
        lib = _lib.load()
        _need_cuda(x1, x2)
        d = x2.shape[-1]
        io = _io_dtype(x2)
        x2f = _flat(x2, d)
        x1f = _flat(x1, d) if gate_mode != GATE_NONE else None
        M = x2f.shape[0]
        out = torch.empty_like(x2f)
        # training: the forward leaves z and gelu'(pre) of its chains (up to 4 x [M, 32*tiles], IO dtype) and the
        # backward neither recomputes the down projections nor reads x1 / x2 for them
        act = None
        if SAVE_ACTIVATIONS and any(ctx.needs_input_grad):
            act = torch.empty(lib.vlpet_saved_bytes(M, pk_a.tiles, io), dtype=torch.uint8, device=x2f.device)
            rc = _timed("k1_fwd", M, lambda: lib.vlpet_adapter_gate_fwd_save(
                _ptr(x1f), x2f.data_ptr(), pk_a.buf.data_ptr(), pk_g.buf.data_ptr() if pk_g is not None else None,
                out.data_ptr(), act.data_ptr(), M, d, pk_a.tiles, gate_mode, float(delta_scale), float(x2_scale),
                float(gate_scale), io, _stream()))
        else:
            rc = _timed("k1_fwd", M, lambda: lib.vlpet_adapter_gate_fwd(
                _ptr(x1f), x2f.data_ptr(), pk_a.buf.data_ptr(), pk_g.buf.data_ptr() if pk_g is not None else None,
                out.data_ptr(), M, d, pk_a.tiles, gate_mode, float(delta_scale), float(x2_scale), float(gate_scale),
                io, _stream()))
        _lib.check(rc, "vlpet_adapter_gate_fwd")
        ctx.act = act
        ctx.link = None
        if link is not None and act is not None and gate_mode != GATE_NONE and x1.requires_grad:
            link.armed = True                   # the tail that follows may park its dx1 for this op's backward
            ctx.link = link
        # the other direction: the sublayer's first GEMM (upstream of x2) armed `out_link` -- this backward parks d/dx1 there
        ctx.out_link = out_link if (out_link is not None and out_link.armed and gate_mode != GATE_NONE) else None
        # the multiplicative gate's backward can start from this output (dq = dy * y * (1 - g): csrc/pet_dz2.hip "YF").  Keeping it is
        # NOT free: the sublayer tail that consumes it saves its own output / pre-norm sum, not y, so one more [M, d] tensor of
        # the IO dtype lives from this forward to its backward -- 12 of them in a BART-base encoder, 24 in T5-base (43 MB each at 28,000
        # rows: +0.5 GB of peak activation memory at the configs[1] batch, measured by bench.py's peak_memory_GB with
        # VLPET_K1_BWD_FROM_X2=1 beside the default; profiles/r06_peak_memory_from_output_ab.txt).  On a 288 GB device the 3 % faster pass 1
        # is the better trade; memory-constrained runs set K1_BWD_FROM_OUTPUT = False.
        ctx.has_y = bool(K1_BWD_FROM_OUTPUT and act is not None and gate_mode == GATE_MUL)
        ctx.save_for_backward(x1f if x1f is not None else x2f, x2f, *params, *((out,) if ctx.has_y else ()))
        ctx.pk = (pk_a, pk_g)
        ctx.cfg = (n_heads, gate_mode, float(delta_scale), float(x2_scale), float(gate_scale), x2.shape,
                   x1.shape if x1 is not None else None)
        return out.view(x2.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x1f, x2f, *params = ctx.saved_tensors
        yf = params.pop() if ctx.has_y else None
        pk_a, pk_g = ctx.pk
        n_heads, gate_mode, sd, s2, gs, shp2, shp1 = ctx.cfg
        M, d = x2f.shape
        io = _io_dtype(x2f)
        gate = gate_mode != GATE_NONE
        dyf = _flat(dy, d)
        dev = x2f.device
        r = pk_a.r
        rg = pk_g.r if gate else 0
        nh2 = 2 * n_heads
        (dwd, s_wd), (dbd, s_bd) = _grad_dest(params[0], (r, d), block=True), _grad_dest(params[n_heads], (r,), block=True)
        (dwu, s_wu), (dbu, s_bu) = _grad_dest(params[nh2], (d, r)), _grad_dest(params[nh2 + 1], (d,))
        dwgd = dbgd = dwgu = dbgu = None
        if gate:
            (dwgd, s_gd), (dbgd, s_gdb) = _grad_dest(params[nh2 + 2], (rg, d)), _grad_dest(params[nh2 + 3], (rg,))
            (dwgu, s_gu), (dbgu, s_gub) = _grad_dest(params[nh2 + 4], (d, rg)), _grad_dest(params[nh2 + 5], (d,))
        dx2 = torch.empty_like(x2f)
        dx1 = torch.empty_like(x2f) if gate else None
        nws = lib.vlpet_bwd_workspace_bytes(M, d, pk_a.tiles, int(gate), io)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        args = (dyf.data_ptr(), _ptr(x1f) if gate else None, x2f.data_ptr(),
                pk_a.buf.data_ptr(), pk_g.buf.data_ptr() if gate else None, _ptr(dx1), dx2.data_ptr(),
                dwd.data_ptr(), dbd.data_ptr(), dwu.data_ptr(), dbu.data_ptr(),
                _ptr(dwgd), _ptr(dbgd), _ptr(dwgu), _ptr(dbgu), r, rg,
                ws.data_ptr(), nws, M, d, pk_a.tiles, gate_mode, sd, s2, gs, io, _stream())
        sinks = [s_wd, s_bd, s_wu, s_bu] + ([s_gd, s_gdb, s_gu, s_gub] if gate else [])
        side = WGRAD_STREAM if all(s is not None for s in sinks) else None
        act = ctx.act

        # the sublayer tail's residual-stream gradient, if it was handed over (ResidualLink): summed inside the kernel
        dx1_in = None
        if ctx.link is not None and ctx.link.dx1 is not None:
            dx1_in, ctx.link.dx1 = ctx.link.dx1, None
            if dx1_in.shape != x2f.shape or dx1_in.dtype != x2f.dtype or not dx1_in.is_contiguous():
                dx1_in = dx1_in.reshape(x2f.shape).to(x2f.dtype).contiguous()
        ctx.link = None

        def phase(ph, a):       # one or both halves of the backward, with or without the forward's saved activations
            if K1_BWD_FINALIZE_LAUNCH:
                ph |= 32            # (A/B: the round-3 form -- partial slabs + a finalize launch -- instead of the in-launch reduce-scatter)
            if act is not None and yf is not None:
                return lib.vlpet_adapter_gate_bwd_saved_y(ph, a[0], a[1], a[2], yf.data_ptr(), act.data_ptr(), a[3], a[4], _ptr(dx1_in),
                                                          *a[5:])
            if act is not None and dx1_in is not None:
                return lib.vlpet_adapter_gate_bwd_saved_acc(ph, a[0], a[1], a[2], act.data_ptr(), a[3], a[4], dx1_in.data_ptr(),
                                                            *a[5:])
            if act is not None:
                return lib.vlpet_adapter_gate_bwd_saved(ph, a[0], a[1], a[2], act.data_ptr(), *a[3:])
            return lib.vlpet_adapter_gate_bwd_phase(ph, *a)

        if side is not None:
            # (bit 2: the form whose first half already delivers dx1 / dx2 -- the two-pass form produces them in its second)
            rc = _timed("k1_bwd_rows", M, lambda: phase(1 | 4, args))
            if rc == 0:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    sargs = args[:-1] + (_stream(),)
                    rc = _timed("k1_bwd_wgrad", M, lambda: phase(2 | 4, sargs))
                for t in (x1f, x2f, dyf, ws, pk_a.buf) + ((pk_g.buf,) if gate else ()) + ((act,) if act is not None else ()) \
                        + ((yf,) if yf is not None else ()):
                    t.record_stream(side)        # the caching allocator must not recycle them under the side stream
        elif K1_BWD_PREVIOUS_SPLIT:
            rc = _timed("k1_bwd_rows", M, lambda: phase(1 | 4, args))
            if rc == 0:
                rc = _timed("k1_bwd_wgrad", M, lambda: phase(2 | 4, args))
        elif (FINALIZE_SIDE_STREAM and gate and act is not None and TIMER is None and all(s is not None for s in sinks)
              and torch.cuda.is_current_stream_capturing() and lib.vlpet_adapter_gate_bwd_form(M, d, pk_a.tiles, io) == 2):
            rc = phase(3 | 8, args)                    # pass 1 + pass 2; the finalize launch as a parallel branch of the captured graph
            if rc == 0:
                rc = _finalize_on_side(lambda st: phase(16, args[:-1] + (st,)), ws)
        elif TIMER is None or not TIMER.wants("k1_bwd_rows"):
            with _deferred_finalize(all(s is not None for s in sinks), ws):
                rc = phase(3, args)
        elif not gate:
            # adapter-only K1 (small / middle gate scripts): its two-pass form is taken for the WHOLE op only (csrc/api.hip ng2), so the
            # bracketed path issues it as one call too -- split phases would time the round-2 row kernel instead of what ships
            rc = TIMER.bracket("k1_bwd_rows", M, lambda: phase(3, args))
        else:       # same work, its kernels bracketed separately
            rc = TIMER.bracket("k1_bwd_rows", M, lambda: phase(1, args))
            two_pass = act is not None and lib.vlpet_adapter_gate_bwd_form(M, d, pk_a.tiles, io) == 2
            if rc == 0 and two_pass:        # (pass 2 and -- where the form has one -- the finalize launch of the column-parallel form)
                rc = TIMER.bracket("k1_bwd_wgrad", M, lambda: phase(2 | 8, args))
                if rc == 0 and (K1_BWD_FINALIZE_LAUNCH or lib.vlpet_adapter_gate_bwd_finalize_launch(M, d, pk_a.tiles, io) == 1):
                    rc = TIMER.bracket("k1_bwd_fin", M, lambda: phase(16, args))
            elif rc == 0:
                rc = TIMER.bracket("k1_bwd_wgrad", M, lambda: phase(2, args))
        ctx.act = None
        _lib.check(rc, "vlpet_adapter_gate_bwd")
        if not gate:
            # without a gate the kernel returns the adapter-branch gradient only; add the residual path
            dx2 = dx2 + s2 * dyf
        rh = r // n_heads
        grads: List[Optional[torch.Tensor]] = []
        if s_wd is not None:
            s_wd.done()
            grads += [None] * n_heads
        else:
            grads += [_grad_like(dwd[i * rh:(i + 1) * rh], params[i]) for i in range(n_heads)]
        if s_bd is not None:
            s_bd.done()
            grads += [None] * n_heads
        else:
            grads += [_grad_like(dbd[i * rh:(i + 1) * rh], params[n_heads + i]) for i in range(n_heads)]
        grads += _finish([(dwu, s_wu, params[nh2]), (dbu, s_bu, params[nh2 + 1])])
        if gate:
            grads += _finish([(dwgd, s_gd, params[nh2 + 2]), (dbgd, s_gdb, params[nh2 + 3]),
                              (dwgu, s_gu, params[nh2 + 4]), (dbgu, s_gub, params[nh2 + 5])])
        gx1 = dx1.view(shp1) if gate else None
        if ctx.out_link is not None:        # the dgrad GEMM of the sublayer's first projection accumulates onto it (_LinearAccFn)
            ctx.out_link.dx1, ctx.out_link.shared = gx1, False
            gx1 = None
            ctx.out_link = None
        return (gx1, dx2.view(shp2), None, None, None, None, None, None, None, None, None, *grads)


def adapter_gate(x1, x2, down_w, down_b, up_w, up_b, gate_params, pk_a: PackedPair, pk_g: Optional[PackedPair],
                 gate_mode=GATE_MUL, delta_scale=1.0, x2_scale=1.0, gate_scale=1.0, link: Optional[ResidualLink] = None,
                 out_link: Optional[ResidualLink] = None):
    """Encoder granularity-controlled adapter (+ low-rank gate).  ``gate_params`` =
    (gate_down_w, gate_down_b, gate_up_w, gate_up_b) or None.  ``link``: see ResidualLink (the tail's d/dx1 comes in);
    ``out_link``: armed by the sublayer's first GEMM (linear_acc) -- this op's d/dx1 goes out through it."""
    params = list(down_w) + list(down_b) + [up_w, up_b]
    if gate_mode != GATE_NONE:
        params += list(gate_params)
    if x2.numel() == 0:
        return _empty_result(x2, params)
    return _AdapterGateFn.apply(x1, x2, pk_a, pk_g, len(down_w), gate_mode, delta_scale, x2_scale, gate_scale, link, out_link,
                                *params)


class _ParallelAdapterFn(torch.autograd.Function):
    """K2: out = y + scale * up(gelu_new(down(x)))."""

    @staticmethod
    def forward(ctx, x, y, pk, scale, link, wd, bd, wu, bu):
        lib = _lib.load()
        _need_cuda(x, y)
        ctx.link = link if (link is not None and link.armed and ctx.needs_input_grad[0]) else None
        d = x.shape[-1]
        io = _io_dtype(x)
        xf, yf = _flat(x, d), _flat(y, d)
        M = xf.shape[0]
        out = torch.empty_like(xf)
        act = None
        if SAVE_ACTIVATIONS and any(ctx.needs_input_grad):      # training: z / gelu'(pre) for the backward (see K1)
            act = torch.empty(lib.vlpet_saved_bytes(M, pk.tiles, io), dtype=torch.uint8, device=xf.device)
            rc = _timed("k2_fwd", M, lambda: lib.vlpet_parallel_adapter_fwd_save(
                xf.data_ptr(), yf.data_ptr(), pk.buf.data_ptr(), out.data_ptr(), act.data_ptr(), M, d, pk.tiles,
                float(scale), io, _stream()))
        else:
            rc = _timed("k2_fwd", M, lambda: lib.vlpet_parallel_adapter_fwd(
                xf.data_ptr(), yf.data_ptr(), pk.buf.data_ptr(), out.data_ptr(), M, d, pk.tiles, float(scale), io, _stream()))
        _lib.check(rc, "vlpet_parallel_adapter_fwd")
        ctx.act = act
        ctx.save_for_backward(xf, wd, bd, wu, bu)
        ctx.pk, ctx.scale, ctx.shape = pk, float(scale), x.shape
        return out.view(y.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xf, wd, bd, wu, bu = ctx.saved_tensors
        pk = ctx.pk
        M, d = xf.shape
        io = _io_dtype(xf)
        dyf = _flat(dy, d)
        f32 = dict(dtype=torch.float32, device=xf.device)
        r = pk.r
        (dwd, s0), (dbd, s1), (dwu, s2) = _grad_dest(wd, (r, d)), _grad_dest(bd, (r,)), _grad_dest(wu, (d, r))
        # bu is None when the up bias belongs to another half of a split bottleneck (encoder_pet._apply_pet_split)
        (dbu, s3) = _grad_dest(bu, (d,)) if bu is not None else (torch.empty(d, dtype=torch.float32, device=xf.device), None)
        dx = torch.empty_like(xf)
        nws = lib.vlpet_bwd_workspace_bytes(M, d, pk.tiles, 0, io)
        ws = torch.empty(nws, dtype=torch.uint8, device=xf.device)
        act = ctx.act
        with _deferred_finalize(s0 is not None and s1 is not None and s2 is not None and (s3 is not None or bu is None), ws, dbu):
            if act is not None:
                rc = _timed("k2_bwd", M, lambda: lib.vlpet_parallel_adapter_bwd_saved(
                    dyf.data_ptr(), xf.data_ptr(), act.data_ptr(), pk.buf.data_ptr(), dx.data_ptr(), dwd.data_ptr(), dbd.data_ptr(),
                    dwu.data_ptr(), dbu.data_ptr(), r, ws.data_ptr(), nws, M, d, pk.tiles, ctx.scale, io, _stream()))
            else:
                rc = _timed("k2_bwd", M, lambda: lib.vlpet_parallel_adapter_bwd(
                    dyf.data_ptr(), xf.data_ptr(), pk.buf.data_ptr(), dx.data_ptr(), dwd.data_ptr(), dbd.data_ptr(),
                    dwu.data_ptr(), dbu.data_ptr(), r, ws.data_ptr(), nws, M, d, pk.tiles, ctx.scale, io, _stream()))
        ctx.act = None
        _lib.check(rc, "vlpet_parallel_adapter_bwd")
        gw = _finish([(dwd, s0, wd), (dbd, s1, bd), (dwu, s2, wu)])
        gbu = _finish([(dbu, s3, bu)])[0] if bu is not None else None
        gx = dx.view(ctx.shape)
        if ctx.link is not None:        # the projection that made `y` from the same `x` accumulates its dgrad onto dx (_LinearAccFn)
            ctx.link.dx1, ctx.link.shared = gx, False
            gx = None
            ctx.link = None
        return (gx, dy, None, None, None, *gw, gbu)


def parallel_adapter(x, y, wd, bd, wu, bu, pk: PackedPair, scale: float = 1.0, link: Optional[ResidualLink] = None):
    """``link``: armed by the frozen projection that produced ``y`` from the same ``x`` (linear_acc); d/dx is parked there."""
    if x.numel() == 0:
        return _empty_result(y, [wd, bd, wu, bu])
    return _ParallelAdapterFn.apply(x, y, pk, scale, link, wd, bd, wu, bu)


LORA_R8_STREAMING = True     # A/B switch (tools/): False = the MFMA kernels at every rank


class _LoraDeltaFn(torch.autograd.Function):
    """K3: out = base + scaling * ((dropout(x) A^T) B^T); base comes from the library GEMM.

    Dropout (lora/controller.py:66): ``p`` > 0 with ``keep`` None -> the kernels' counter-based generator keyed by
    ``seed`` (forward, backward rows kernel and weight-gradient kernel regenerate the same mask; nothing is stored);
    ``keep`` given -> that explicit 0/1 mask.  ``want_mask``: also return the applied mask (parity tests)."""

    @staticmethod
    def forward(ctx, x, base, pk, scaling, keep, p, seed, want_mask, lora_a, lora_b, link=None):
        # (`link`: armed by the frozen projection that produced `base` from the same x -- its dgrad GEMM takes over d/dx)
        ctx.link = link if (link is not None and link.armed and ctx.needs_input_grad[0]) else None
        lib = _lib.load()
        _need_cuda(x, base)
        d = x.shape[-1]
        io = _io_dtype(x)
        xf, bf = _flat(x, d), _flat(base, d)
        M = xf.shape[0]
        out = torch.empty_like(bf)
        kf = None
        if keep is not None:
            kf = _flat(keep, d)
            if kf.dtype != torch.uint8:
                kf = kf.to(torch.uint8)
        mask = torch.empty(M, d, dtype=torch.uint8, device=xf.device) if (want_mask and p > 0) else None
        act = None
        train_form = SAVE_ACTIVATIONS and any(ctx.needs_input_grad)
        if LORA_R8_STREAMING and lib.vlpet_lora_r8_applies(M, d, pk.r, io):
            # rank <= 8: the streaming row kernel (csrc/lora8.hip) -- same pack, same masks, same saved block as the MFMA form
            if train_form:
                act = torch.empty(lib.vlpet_lora_saved_bytes(M, d, pk.tiles, io), dtype=torch.uint8, device=xf.device)
            rc = _timed("k3_fwd", M, lambda: lib.vlpet_lora_delta_fwd_r8(
                xf.data_ptr(), bf.data_ptr(), pk.buf.data_ptr(), _ptr(kf), float(p), int(seed), _ptr(mask), out.data_ptr(),
                _ptr(act), M, d, pk.r, float(scaling), io, _stream()))
        elif train_form:      # training: z = dropout(x) A^T for the backward (see K1 / K2)
            act = torch.empty(lib.vlpet_lora_saved_bytes(M, d, pk.tiles, io), dtype=torch.uint8, device=xf.device)
            rc = _timed("k3_fwd", M, lambda: lib.vlpet_lora_delta_fwd_save(
                xf.data_ptr(), bf.data_ptr(), pk.buf.data_ptr(), _ptr(kf), float(p), int(seed), _ptr(mask), out.data_ptr(),
                act.data_ptr(), M, d, pk.tiles, float(scaling), io, _stream()))
        else:
            rc = _timed("k3_fwd", M, lambda: lib.vlpet_lora_delta_fwd(
                xf.data_ptr(), bf.data_ptr(), pk.buf.data_ptr(), _ptr(kf), float(p), int(seed), _ptr(mask), out.data_ptr(),
                M, d, pk.tiles, float(scaling), io, _stream()))
        _lib.check(rc, "vlpet_lora_delta_fwd")
        ctx.act = act
        ctx.save_for_backward(xf, lora_a, lora_b)
        ctx.keep = kf
        ctx.cfg = (pk, float(scaling), float(p), int(seed), x.shape)
        out = out.view(base.shape)
        if want_mask:
            if mask is None:
                mask = torch.ones(M, d, dtype=torch.uint8, device=xf.device)
            ctx.mark_non_differentiable(mask)
            return out, mask.view(x.shape)
        return out

    @staticmethod
    def backward(ctx, dy, *unused):
        lib = _lib.load()
        xf, lora_a, lora_b = ctx.saved_tensors
        pk, scaling, p, seed, shape = ctx.cfg
        M, d = xf.shape
        io = _io_dtype(xf)
        dyf = _flat(dy, d)
        r = pk.r
        (da, s0), (db, s1) = _grad_dest(lora_a, (r, d)), _grad_dest(lora_b, (d, r))
        dx = torch.empty_like(xf)
        nws = lib.vlpet_bwd_workspace_bytes(M, d, pk.tiles, 0, io)
        ws = torch.empty(nws, dtype=torch.uint8, device=xf.device)
        act = ctx.act
        with _deferred_finalize(s0 is not None and s1 is not None, ws):
            if act is not None:
                rc = _timed("k3_bwd", M, lambda: lib.vlpet_lora_delta_bwd_saved(
                    dyf.data_ptr(), xf.data_ptr(), act.data_ptr(), pk.buf.data_ptr(), _ptr(ctx.keep), p, seed, dx.data_ptr(),
                    da.data_ptr(), db.data_ptr(), r, ws.data_ptr(), nws, M, d, pk.tiles, scaling, io, _stream()))
            else:
                rc = _timed("k3_bwd", M, lambda: lib.vlpet_lora_delta_bwd(
                    dyf.data_ptr(), xf.data_ptr(), pk.buf.data_ptr(), _ptr(ctx.keep), p, seed, dx.data_ptr(), da.data_ptr(),
                    db.data_ptr(), r, ws.data_ptr(), nws, M, d, pk.tiles, scaling, io, _stream()))
        ctx.act = None
        _lib.check(rc, "vlpet_lora_delta_bwd")
        gx = dx.view(shape)
        if ctx.link is not None:
            ctx.link.dx1, ctx.link.shared = gx, False
            gx = None
            ctx.link = None
        return (gx, dy, None, None, None, None, None, None, *_finish([(da, s0, lora_a), (db, s1, lora_b)]), None)


def lora_delta(x, base, lora_a, lora_b, pk: PackedPair, scaling: float, keep=None, p: float = 0.0, seed: int = 0,
               return_mask: bool = False, link: Optional["ResidualLink"] = None):
    """``base + scaling * (dropout(x, p) @ A^T @ B^T)``.  ``keep`` (uint8 [.., d]) overrides the generator's mask."""
    if x.numel() == 0:
        out = _empty_result(base, [lora_a, lora_b])
        return (out, torch.empty(x.shape, dtype=torch.uint8, device=x.device)) if return_mask else out
    return _LoraDeltaFn.apply(x, base, pk, scaling, keep, p, seed, return_mask, lora_a, lora_b, link)
