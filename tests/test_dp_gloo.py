"""Data-parallel gradient exchange on CPU (gloo, world_size 2): N ranks with 1/N of the batch each
must produce the same averaged trainable gradients and the same post-step parameters as one rank with
the whole batch (the reference offers no working multi-GPU oracle)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vlpet_amd.train as TR


def _model():
    torch.manual_seed(7)
    m = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                            torch.nn.Linear(32, 4))
    for p in m[2].parameters():      # a frozen middle: only trainable grads travel
        p.requires_grad = False
    return m


def _data():
    g = torch.Generator().manual_seed(3)
    return torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    fg = TR.FlatGrads(m, world_size=world, n_buckets=2)
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2)
    x, y = _data()
    xs, ys = x.chunk(world)[rank], y.chunk(world)[rank]
    for _ in range(3):
        fg.zero()
        ((m(xs) - ys) ** 2).mean().backward()
        fg.finish()
        fg.clip_(5.0)
        opt.step()
    if rank == 0:
        torch.save(dict(flat=fg.flat.clone(), params=[p.detach().clone() for p in fg.params]), out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    m = _model()
    fg = TR.FlatGrads(m, world_size=1)
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2)
    x, y = _data()
    for _ in range(3):
        fg.zero()
        ((m(x) - y) ** 2).mean().backward()
        fg.finish()
        fg.clip_(5.0)
        opt.step()
    torch.testing.assert_close(got["flat"], fg.flat, rtol=1e-5, atol=1e-6)
    for a, b in zip(got["params"], fg.params):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


# ---- the direct-write gradient sinks (train.GradSink) under data parallelism ---------------------------------------
class _SinkLinearFn(torch.autograd.Function):
    """Stands in for the HIP weight-gradient kernels: writes dW / db straight into the parameter's slot of the flat
    buffer when the trainer offers one (first write of the step) and returns None to autograd, exactly like
    functional._grad_dest / _finish do around the real kernels."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        import vlpet_amd.functional as VF
        x, w, b = ctx.saved_tensors
        (dw, sw), (db, sb) = VF._grad_dest(w, w.shape), VF._grad_dest(b, b.shape)
        dw.copy_(dy.t() @ x)
        db.copy_(dy.sum(0))
        gw, gb = VF._finish([(dw, sw, w), (db, sb, b)])
        return dy @ w, gw, gb


class _SinkNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(11)
        self.a = torch.nn.Linear(16, 24)      # gradients through the sinks
        self.mid = torch.nn.Linear(24, 24)    # ordinary autograd accumulation (post-accumulate hooks)
        self.b = torch.nn.Linear(24, 4)       # sinks again
        self.twice = False                    # use b twice per step (second use must fall back to autograd; world 1 only)

    def forward(self, x):
        h = torch.tanh(_SinkLinearFn.apply(x, self.a.weight, self.a.bias))
        h = torch.tanh(self.mid(h))
        y = _SinkLinearFn.apply(h, self.b.weight, self.b.bias)
        if self.twice:
            y = y + 0.5 * _SinkLinearFn.apply(h * h, self.b.weight, self.b.bias)
        return y


def _sink_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _SinkNet()
    fg = TR.FlatGrads(m, world_size=world, n_buckets=3, flatten_params=True, sinks=True)
    x, y = _data()
    xs, ys = x.chunk(world)[rank], y.chunk(world)[rank]
    launches = []
    orig = fg._launch
    fg._launch = lambda b: (launches.append(b), orig(b))[1]
    res = []
    for _ in range(2):
        fg.begin_step(zero=True)
        ((m(xs) - ys) ** 2).mean().backward()
        fg.finish(average=True)
        res.append(fg.flat.clone())
    if rank == 0:
        taken = sum(1 for p in fg.params for s in (getattr(p, "_vlpet_sink", None),) if s is not None and s.epoch == fg.epoch)
        torch.save(dict(flats=res, launches=launches, taken=taken, nb=len(fg.buckets)), out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_sinks_under_data_parallel(tmp_path):
    """Two gloo ranks with half the batch each: gradients written through the sinks and gradients accumulated by
    autograd arrive averaged in the flat buffer; every bucket is all-reduced exactly once per step."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sink.pt")
    mp.spawn(_sink_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["taken"] == 4                                   # a.weight, a.bias, b.weight, b.bias took the direct path
    assert sorted(got["launches"]) == sorted(list(range(got["nb"])) * 2)
    # one rank, whole batch, plain autograd
    m = _SinkNet()
    x, y = _data()
    ref = TR.FlatGrads(m, world_size=1, sinks=False)
    ref.begin_step(zero=True)
    ((m(x) - y) ** 2).mean().backward()
    for f in got["flats"]:
        torch.testing.assert_close(f, ref.flat, rtol=1e-5, atol=1e-6)


def test_gradient_sink_second_use_falls_back_to_autograd():
    """One process: a parameter used twice in a step takes the direct-write path once and autograd accumulation for
    the second contribution; the flat buffer holds the sum.  (Under data parallelism the same model is rejected.)"""
    m = _SinkNet(); m.twice = True
    x, y = _data()
    fg = TR.FlatGrads(m, world_size=1, flatten_params=True, sinks=True)
    fg.begin_step(zero=True)
    ((m(x) - y) ** 2).mean().backward()
    ref_m = _SinkNet(); ref_m.twice = True
    ref = TR.FlatGrads(ref_m, world_size=1, sinks=False)
    ref.begin_step(zero=True)
    ((ref_m(x) - y) ** 2).mean().backward()
    torch.testing.assert_close(fg.flat, ref.flat, rtol=1e-5, atol=1e-6)
    fg.dp = True                               # pretend DP: the second use must be refused, not silently raced
    fg.begin_step(zero=True)
    with pytest.raises(RuntimeError):
        ((m(x) - y) ** 2).mean().backward()

