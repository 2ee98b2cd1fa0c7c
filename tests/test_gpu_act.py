"""FFN activation + dropout pass (csrc/actdrop.hip, vlpet_amd.act) against the eager pair it replaces
(my_transformers/modeling_bart.py:1264-1265: activation_fn(fc1(x)) then F.dropout; T5DenseReluDense: relu then dropout),
with the mask the kernel applied exported so that forward and backward can be compared element for element."""
import pytest
import torch
import torch.nn.functional as F

from oracle import vlpet_oracle as O
from gpu_cases import rel_err

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 1e-2}
EAGER = {"gelu": F.gelu, "gelu_new": O.gelu_new, "relu": F.relu}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu", "gelu_new", "relu"])
@pytest.mark.parametrize("p", [0.0, 0.1, 0.5])
def test_act_dropout_matches_the_eager_pair(dtype, act, p):
    from vlpet_amd.act import act_dropout
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(7, 33, 3072, generator=g) * 2.0).to(dtype)
    dy = torch.randn(7, 33, 3072, generator=g).to(dtype)
    xg = x.cuda().requires_grad_(True)
    out, keep = act_dropout(xg, act, p, True, seed=1234, return_mask=True)
    out.backward(dy.cuda())
    keep = keep.cpu()
    assert keep.dtype == torch.uint8 and set(keep.unique().tolist()) <= {0, 1}
    if p == 0.0:
        assert int(keep.min()) == 1
    else:
        frac = float(keep.float().mean())
        assert abs(frac - (1.0 - p)) < 5e-3, frac                  # 710 k elements: 3 sigma is ~2e-3
    xr = x.float().requires_grad_(True)
    ref = EAGER[act](xr) * keep.float() / (1.0 - p)
    ref.backward(dy.float())
    assert rel_err(out.float().cpu(), ref.detach()) <= TOL[dtype]
    assert rel_err(xg.grad.float().cpu(), xr.grad) <= TOL[dtype]


def test_act_dropout_mask_is_a_function_of_the_seed_only():
    from vlpet_amd.act import act_dropout
    x = torch.randn(64, 3072, device="cuda", dtype=torch.bfloat16)
    _, k1 = act_dropout(x, "gelu", 0.1, True, seed=7, return_mask=True)
    _, k2 = act_dropout(x.float(), "relu", 0.1, True, seed=7, return_mask=True)     # other dtype, other activation: same mask
    _, k3 = act_dropout(x, "gelu", 0.1, True, seed=8, return_mask=True)
    assert torch.equal(k1, k2)
    assert not torch.equal(k1, k3)
    o_eval = act_dropout(x, "gelu", 0.1, False)                                      # eval: no dropout whatever p says
    assert rel_err(o_eval.float().cpu(), F.gelu(x.float()).cpu()) <= 1e-2


def test_act_dropout_argument_errors():
    from vlpet_amd import _lib
    from vlpet_amd.act import act_dropout
    with pytest.raises(RuntimeError):
        act_dropout(torch.randn(4, 16), "gelu", 0.0, False)                         # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        act_dropout(torch.randn(4, 12, device="cuda"), "gelu", 0.0, False)          # last dim not a multiple of 8
    with pytest.raises(RuntimeError):
        act_dropout(torch.randn(4, 16, device="cuda"), "swish", 0.0, False)
    lib = _lib.load()
    x = torch.randn(64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vlpet_act_dropout_fwd(x.data_ptr(), x.data_ptr(), None, 60, 0, 0.0, 0, _lib.VLPET_F32, st) == -1     # n % 8
    assert lib.vlpet_act_dropout_fwd(x.data_ptr(), x.data_ptr(), None, 64, 9, 0.0, 0, _lib.VLPET_F32, st) == -1     # act
    assert lib.vlpet_act_dropout_fwd(x.data_ptr(), x.data_ptr(), None, 64, 0, 1.0, 0, _lib.VLPET_F32, st) == -1     # p
    assert lib.vlpet_act_dropout_fwd(None, x.data_ptr(), None, 64, 0, 0.0, 0, _lib.VLPET_F32, st) == -5
    assert lib.vlpet_act_dropout_bwd(x.data_ptr(), x.data_ptr() + 4, x.data_ptr(), 8, 0, 0.0, 0, _lib.VLPET_F32, st) == -3


def test_host_encoder_layer_uses_the_fused_activation(monkeypatch):
    """The BART host layer goes through the HIP pass (not F.gelu + F.dropout) and agrees with the eager pair at p = 0."""
    import vlpet_amd.host.bart as HB
    calls = []
    real = HB.ffn_activation
    monkeypatch.setattr(HB, "ffn_activation", lambda x, act, p, tr: (calls.append((act, p, tr)), real(x, act, p, tr))[1])
    cfg = HB.vlpet_config(d_model=64, encoder_attention_heads=4, encoder_ffn_dim=128, adapter_down_dim=8, adapter_gating_down_dim=8,
                          dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    torch.manual_seed(0)
    layer = HB.BartEncoderLayer(cfg).cuda().eval()
    x = torch.randn(2, 8, 64, device="cuda")
    y = layer(x)
    assert calls == [("gelu", 0.0, False)]
    monkeypatch.setattr(HB, "ffn_activation", lambda x, act, p, tr: F.dropout(F.gelu(x), p=p, training=tr))
    y_ref = layer(x)
    assert rel_err(y, y_ref) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,La,Lv,d,p", [(5, 20, 36, 768, 0.1), (3, 7, 72, 64, 0.3), (4, 40, 36, 768, 0.0), (500, 20, 36, 768, 0.1)])
def test_concat_dropout_is_cat_then_dropout_with_the_regenerated_mask(B, La, Lv, d, p, dtype):
    """act.concat_dropout (src/modeling_bart.py:804-820): x = dropout(cat([a, v], 1)); every element is either 0 or its source / (1 - p),
    the kept fraction is 1 - p, and the backward applies the SAME mask to the two slices of the gradient."""
    from vlpet_amd.act import concat_dropout
    torch.manual_seed(0)
    a = (torch.randn(B, La, d) + 3.0).cuda().to(dtype).requires_grad_(True)       # (bounded away from 0: a zero output is a dropped element)
    v = (torch.randn(B, Lv, d) - 3.0).cuda().to(dtype).requires_grad_(True)
    x = concat_dropout(a, v, p, training=True, seed=11)
    ref = torch.cat([a, v], 1).detach().float()
    keep = x.detach().float() != 0
    if p == 0.0:
        assert torch.equal(x.detach(), torch.cat([a, v], 1).detach())
    else:
        frac = float(keep.float().mean())
        assert abs(frac - (1 - p)) < 0.02, frac
        scaled = (ref / (1 - p)).to(dtype).float()
        assert float(((x.detach().float() - scaled) * keep).abs().max()) <= 2.0 ** -7 * float(scaled.abs().max())
    dx = torch.randn(B, La + Lv, d).cuda().to(dtype)
    x.backward(dx)
    want = (dx.float() * keep / (1 - p)).to(dtype).float()
    assert float((a.grad.float() - want[:, :La]).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    assert float((v.grad.float() - want[:, La:]).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    # the same seed gives the same mask; eval mode is the plain concatenation
    x2 = concat_dropout(a, v, p, training=True, seed=11)
    assert torch.equal(x2, x)
    assert torch.equal(concat_dropout(a, v, p, training=False), torch.cat([a, v], 1))
    # only one side wants a gradient
    v2 = v.detach()
    x3 = concat_dropout(a, v2, p, training=True, seed=11)
    a.grad = None
    x3.backward(dx)
    assert float((a.grad.float() - want[:, :La]).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
