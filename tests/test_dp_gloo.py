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
