"""GPU: vlpet_sum_n and functional.fanout -- the gradient of a tensor that several consumers read (the encoder output under every
decoder layer's cross-attention, my_transformers/modeling_bart.py:2300-2330) summed in ONE launch instead of autograd's pairwise adds."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n", [1, 2, 3, 6, 8, 9, 12, 17])
def test_sum_n_through_the_abi(n, dtype):
    from vlpet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(n)
    L = 8 * 12345                       # not a multiple of the block size
    srcs = [torch.randn(L, device="cuda").to(dtype) for _ in range(n)]
    ref = torch.stack([s.float() for s in srcs]).sum(0)
    out = torch.empty(L, dtype=dtype, device="cuda")
    arr = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    io = _lib.VLPET_F32 if dtype == torch.float32 else _lib.VLPET_BF16
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vlpet_sum_n(arr, n, out.data_ptr(), L, io, st) == 0
    torch.cuda.synchronize()
    # one rounding per launch of eight sources: n <= 8 is the exactly rounded fp32 sum; beyond, a rounding per extra launch
    tol = (2.0 ** -8 if dtype == torch.bfloat16 else 1e-6) * (1 + (n - 1) // 7)
    assert float((out.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    if n <= 8 and dtype == torch.bfloat16:
        assert torch.equal(out, ref.to(dtype))
    # in place over the first source
    assert lib.vlpet_sum_n(arr, n, srcs[0].data_ptr(), L, io, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(srcs[0], out)
    assert lib.vlpet_sum_n(arr, n, out.data_ptr(), L + 4, io, st) == -1        # VLPET_E_SHAPE: len % 8


def test_fanout_sums_the_consumers_gradients_in_one_launch(monkeypatch):
    import vlpet_amd.functional as VF
    torch.manual_seed(0)
    x = torch.randn(7, 33, 768, device="cuda").bfloat16().requires_grad_(True)
    ws = [torch.randn(768, 768, device="cuda").bfloat16() * 0.05 for _ in range(6)]
    calls = []
    orig = VF._FanoutFn.backward
    monkeypatch.setattr(VF._FanoutFn, "backward", staticmethod(lambda ctx, *gs: (calls.append(len(gs)), orig(ctx, *gs))[1]))
    ys = VF.fanout(x, 6)
    assert all(y.data_ptr() == x.data_ptr() for y in ys)
    sum((y @ w).float().square().sum() for y, w in zip(ys, ws)).backward()
    g_fan, x.grad = x.grad, None
    assert calls == [6]
    sum((x @ w).float().square().sum() for w in ws).backward()        # autograd's pairwise accumulation
    ref64 = sum(((2 * (x.detach().double() @ w.double())) @ w.double().t()) for w in ws)
    e_fan = float((g_fan.double() - ref64).abs().max()); e_auto = float((x.grad.double() - ref64).abs().max())
    assert e_fan <= 2.0 ** -7 * float(ref64.abs().max()) and e_fan <= 1.5 * e_auto + 1e-6       # one rounding instead of five
    # off the GPU path (no gradient wanted / switch off): plain references
    with torch.no_grad():
        assert all(y is x for y in VF.fanout(x, 3))
    monkeypatch.setattr(VF, "FANOUT_SUM", False)
    assert all(y is x for y in VF.fanout(x, 3))
