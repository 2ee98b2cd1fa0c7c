"""LM-head cross entropy (csrc/celoss.hip, vlpet_amd.lmloss) against torch's CrossEntropyLoss(ignore_index=-100,
reduction='none') on the same logits (src/modeling_bart.py:1583-1586), forward and backward, and the head + loss
composition against the eager chain it replaces (:1574: lm_head(h) + final_logits_bias)."""
import pytest
import torch
import torch.nn.functional as F

from gpu_cases import rel_err

pytestmark = pytest.mark.gpu


def _case(N, V, dtype, seed=0, scale=4.0):
    g = torch.Generator().manual_seed(seed)
    ld = (V + 7) // 8 * 8
    logits = torch.zeros(N, ld)
    logits[:, :V] = torch.randn(N, V, generator=g) * scale
    logits[:, V:] = 50.0                                   # padding columns hold junk: they must not enter the softmax
    labels = torch.randint(0, V, (N,), generator=g)
    labels[::5] = -100                                     # ignored tokens
    labels[1] = V - 1                                      # last valid column
    labels[2] = 0
    dloss = torch.randn(N, generator=g)
    dloss[3] = 0.0
    return logits.to(dtype), labels, dloss, ld


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-6), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("N,V", [(37, 50465), (6, 32200), (129, 1003), (4, 8), (5, 5)])
def test_cross_entropy_rows_matches_torch(N, V, dtype, tol):
    from vlpet_amd.lmloss import cross_entropy_rows
    logits, labels, dloss, ld = _case(N, V, dtype, seed=N + V)
    lg = logits.cuda().requires_grad_(True)
    loss = cross_entropy_rows(lg, labels.cuda(), V)
    assert loss.dtype == torch.float32 and loss.shape == (N,)
    loss.backward(dloss.cuda())
    ref_in = logits[:, :V].float().requires_grad_(True)
    ref = F.cross_entropy(ref_in, labels, ignore_index=-100, reduction="none")
    ref.backward(dloss)
    assert rel_err(loss, ref) <= 2e-6                      # fp32 arithmetic on the same (already rounded) logits
    got = lg.grad.float().cpu()
    assert float(got[:, V:].abs().max()) == 0.0 if ld > V else True            # padding columns: exact zeros
    assert float(got[labels < 0].abs().max()) == 0.0                           # ignored rows: exact zeros
    assert float(got[3].abs().max()) == 0.0                                    # zero weight row
    assert rel_err(got[:, :V], ref_in.grad) <= tol


def test_large_logits_do_not_overflow():
    from vlpet_amd.lmloss import cross_entropy_rows
    x = torch.full((4, 64), -3.0e4)
    x[:, 7] = 8.0e4
    x[2, 9] = 8.0e4
    labels = torch.tensor([7, 3, 9, -100])
    loss = cross_entropy_rows(x.cuda(), labels.cuda())
    ref = F.cross_entropy(x, labels, ignore_index=-100, reduction="none")
    assert torch.isfinite(loss).all()
    assert rel_err(loss, ref) <= 1e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("with_bias,trainable", [(False, False), (True, False), (False, True)])
def test_lm_head_loss_matches_the_eager_chain(dtype, tol, with_bias, trainable):
    from vlpet_amd.lmloss import lm_head_loss
    g = torch.Generator().manual_seed(11)
    B, T, d, V = 5, 7, 64, 1003
    h = torch.randn(B, T, d, generator=g).to(dtype)
    W = (torch.randn(V, d, generator=g) * 0.2)
    bias = torch.randn(1, V, generator=g) * 0.5 if with_bias else torch.zeros(1, V)
    labels = torch.randint(0, V, (B, T), generator=g)
    labels[:, -2:] = -100
    wts = torch.rand(B, T, generator=g)
    hg = h.cuda().requires_grad_(True)
    Wg = W.cuda().requires_grad_(trainable)
    loss, logits = lm_head_loss(hg, Wg, labels.cuda(), bias.cuda())
    assert loss.shape == labels.shape and logits.shape == (B, T, V)
    (loss * wts.cuda()).sum().backward()
    hr = h.float().requires_grad_(True)
    Wr = W.to(dtype).float().requires_grad_(trainable)
    lr = F.linear(hr, Wr) + bias
    ref = F.cross_entropy(lr.view(-1, V), labels.view(-1), ignore_index=-100, reduction="none").view(B, T)
    (ref * wts).sum().backward()
    assert rel_err(logits, lr) <= tol
    assert rel_err(loss, ref) <= tol
    assert rel_err(hg.grad, hr.grad) <= tol
    if trainable:
        assert rel_err(Wg.grad, Wr.grad) <= tol
    # frozen head: the padded copy is cached and refreshed when the weight is edited in place
    if not trainable:
        from vlpet_amd import lmloss
        first = lmloss._padded_head(Wg, dtype)
        assert lmloss._padded_head(Wg, dtype) is first
        Wg.mul_(2.0)
        assert lmloss._padded_head(Wg, dtype) is not first


def test_cross_entropy_argument_errors():
    from vlpet_amd import _lib
    from vlpet_amd.lmloss import cross_entropy_rows
    with pytest.raises(RuntimeError):
        cross_entropy_rows(torch.randn(4, 16), torch.zeros(4, dtype=torch.long))             # CPU tensors
    with pytest.raises(RuntimeError):
        cross_entropy_rows(torch.randn(4, 12, device="cuda"), torch.zeros(4, dtype=torch.long, device="cuda"))
    lib = _lib.load()
    x = torch.randn(4, 16, device="cuda"); lab = torch.zeros(4, dtype=torch.long, device="cuda")
    out = torch.empty(4, device="cuda"); st = torch.cuda.current_stream().cuda_stream
    assert lib.vlpet_ce_loss_fwd(x.data_ptr(), lab.data_ptr(), out.data_ptr(), out.data_ptr(), 4, 17, 16, _lib.VLPET_F32, st) == -1   # V > ld
    assert lib.vlpet_ce_loss_fwd(x.data_ptr(), lab.data_ptr(), out.data_ptr(), out.data_ptr(), 4, 12, 12, _lib.VLPET_F32, st) == -1   # ld % 8
    assert lib.vlpet_ce_loss_fwd(x.data_ptr(), None, out.data_ptr(), out.data_ptr(), 4, 16, 16, _lib.VLPET_F32, st) == -5
    assert lib.vlpet_ce_loss_fwd(x.data_ptr(), lab.data_ptr(), out.data_ptr(), out.data_ptr(), 4, 16, 16, 7, st) == -6


def test_out_of_range_label_raises_and_empty_batch_is_empty():
    """a label outside [0, V) other than -100 used to be treated like ignore_index (F.cross_entropy asserts);
    the first batch per vocabulary is now validated on the host.  N == 0 returns an empty loss instead of a shape error."""
    import vlpet_amd.lmloss as L
    V = 40
    lg = torch.randn(6, V, device="cuda", dtype=torch.bfloat16)
    old = L.CHECK_LABELS
    L.CHECK_LABELS = "always"
    try:
        with pytest.raises(IndexError):
            L.cross_entropy_rows(lg, torch.tensor([1, 2, V, 3, -100, 0], device="cuda"), V)
        with pytest.raises(IndexError):
            L.cross_entropy_rows(lg, torch.tensor([1, 2, -7, 3, -100, 0], device="cuda"), V)
        ok = L.cross_entropy_rows(lg, torch.tensor([1, 2, V - 1, 3, -100, 0], device="cuda"), V)
        assert ok.shape == (6,) and float(ok[4]) == 0.0
    finally:
        L.CHECK_LABELS = old
    e = L.cross_entropy_rows(lg[:0].requires_grad_(True), torch.zeros(0, dtype=torch.long, device="cuda"), V)
    assert e.shape == (0,) and e.dtype == torch.float32
    e.sum().backward()


def test_later_batches_with_bad_labels_are_tallied_on_the_device():
    """the host-side check looks at the first batch per vocabulary only.  Every batch is counted by the forward kernel
    (vlpet_ce_loss_fwd_checked: labels outside [0, V) other than -100), without a synchronisation; lmloss.bad_label_count reads the
    tally and train.Trainer.check_labels raises on it."""
    import vlpet_amd.lmloss as L
    V = 48
    lg = torch.randn(8, V, device="cuda", dtype=torch.bfloat16)
    old = L.CHECK_LABELS
    L.CHECK_LABELS = "never"                                     # (as if this vocabulary's first batch had been clean)
    try:
        before = L.bad_label_count()
        ok = L.cross_entropy_rows(lg, torch.tensor([1, 2, 3, -100, 0, 5, 6, 7], device="cuda"), V)
        assert L.bad_label_count() == before and float(ok[3]) == 0.0
        out = L.cross_entropy_rows(lg, torch.tensor([1, V, 3, -100, -5, 5, V + 9, 7], device="cuda"), V)
        assert L.bad_label_count() == before + 3                 # V, -5, V + 9; -100 is the ignore index
        assert float(out[1]) == 0.0 and float(out[4]) == 0.0 and float(out[6]) == 0.0 and float(out[0]) > 0.0
    finally:
        L.CHECK_LABELS = old
        for t in L._BAD.values():                                # leave a clean tally for the trainers of later tests
            t.zero_()
