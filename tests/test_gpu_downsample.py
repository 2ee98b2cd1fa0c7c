"""Downsample HIP kernel (csrc/downsample.hip) against the reference-generated fixture and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import vlpet_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_downsample_fixture_bit_exact():
    from vlpet_amd.visual import Downsample
    g = np.load(os.path.join(G, "downsample_7to6_d64.npz"))
    t = lambda k: torch.from_numpy(g[k]).cuda()
    ds = Downsample((6, 6))
    y, yb = ds((t("x"), t("boxes")))
    assert torch.equal(y.cpu(), torch.from_numpy(g["y"])) and torch.equal(yb.cpu(), torch.from_numpy(g["yb"]))
    y2, b2, i2, o2 = ds((t("x2"), t("boxes2"), t("img_ids"), t("obj_ids")))
    assert torch.equal(y2.cpu(), torch.from_numpy(g["y2"])) and torch.equal(b2.cpu(), torch.from_numpy(g["yb2"]))
    assert torch.equal(i2.cpu(), torch.from_numpy(g["yi2"])) and torch.equal(o2.cpu(), torch.from_numpy(g["yo2"]))


@pytest.mark.parametrize("B,L,dim,out", [(3, 49, 2048, 6), (1, 49, 8, 6), (2, 64, 512, 4), (2, 49, 2048, 7), (5, 100, 72, 3)])
def test_downsample_matches_oracle(B, L, dim, out):
    from vlpet_amd.visual import Downsample
    gen = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, L, dim, generator=gen)
    ds = Downsample((out, out))
    ref = O.downsample(x, (out, out))
    assert torch.equal(ds.downsample_inputs(x.cuda()).cpu(), ref)                       # fp32 -> fp32: bit exact
    yb = ds.downsample_inputs(x.cuda(), out_dtype=torch.bfloat16).cpu()                 # fused cast == cast of the pool
    assert torch.equal(yb, ref.to(torch.bfloat16))
    yb2 = ds.downsample_inputs(x.cuda().bfloat16()).cpu()                               # bf16 in
    assert torch.equal(yb2, O.downsample(x.bfloat16().float(), (out, out)).to(torch.bfloat16))


def test_downsample_bench_shape_nlvr():
    """BASELINE configs[1] NLVR shape: [166, 98, 2048] fp32 -> [166, 72, 2048] bf16."""
    from vlpet_amd.visual import Downsample
    gen = torch.Generator().manual_seed(3)
    B = 166
    x = torch.randn(B, 98, 2048, generator=gen)
    boxes = torch.zeros(B, 98, 4)
    img = torch.cat([torch.zeros(B, 49), torch.ones(B, 49)], 1).long()
    obj = torch.cat([torch.arange(49), torch.arange(49)]).unsqueeze(0).expand(B, -1).contiguous()
    y, b, i, o = Downsample((6, 6))((x.cuda(), boxes.cuda(), img.cuda(), obj.cuda()), out_dtype=torch.bfloat16)
    yr, br, ir, orr = O.downsample_nlvr(x, boxes, img, obj)
    assert torch.equal(y.cpu(), yr.to(torch.bfloat16)) and torch.equal(i.cpu(), ir) and torch.equal(o.cpu(), orr)
    assert y.shape == (B, 72, 2048) and b.shape == (B, 72, 4)
