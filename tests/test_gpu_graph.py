"""train.Trainer(graph=True): forward + backward of a step captured once per batch signature with hipGraph and replayed
(multitask.py:217-342 is the step it stands for; the reference has no graph mode -- it is how the strong-scaled per-rank batches
of an 8-GPU run stop being bound by ~2,000 host-side launches per step).  Checked here: (i) without dropout a replayed trainer
lands on the SAME losses and parameters as the eager trainer over steps that change the weights -- i.e. every weight-derived
buffer (fragment packs, K4 pack, IO-dtype copies) is refreshed without the Python forward; (ii) with dropout the masks of the
dropout-carrying kernels change from step to step under replay (the device step counter of vlpet_set_seed_counter) and are
the same function of (seed, counter) forward and backward; (iii) two ranks with deferred gradient exchange equal one."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny(dropout=0.0, **kw):
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    cfg = HB.vlpet_config(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=500,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout=dropout, attention_dropout=dropout,
                          activation_dropout=dropout, **kw)
    torch.manual_seed(0)
    model = HB.VLBart(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    model.train()
    return model, cfg


def _cuda_batch(b):
    bb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    bb["vis_inputs"] = tuple(t.cuda() for t in b["vis_inputs"])
    return bb


@pytest.fixture(autouse=True)
def _no_seed_counter_left_behind():
    yield
    from vlpet_amd import _lib
    _lib.load().vlpet_set_seed_counter(None)


@pytest.mark.parametrize("lora", [False, True])
def test_replayed_trainer_equals_eager_trainer(lora):
    import vlpet_amd.train as TR
    kw = dict(use_adapter=False, use_encoder_adapter_down_multihead=False, use_encoder_adapter_gating_large_x_lowrank=False,
              use_decoder_enc_attn_value_parallel_adapter_down_dim=False, unfreeze_encoder_layer_norms=False,
              use_lora=True, lora_dim=8, use_single_lora=True, lora_dropout=0.0) if lora else {}
    model, cfg = _tiny(0.0, **kw)
    m_eager, m_graph = copy.deepcopy(model).cuda(), copy.deepcopy(model).cuda()
    gen = torch.Generator().manual_seed(5)
    order = ("vqa", "nlvr", "caption") * 4                    # per shape: eager, capture + replay, replay, replay
    fixed = {t: _cuda_batch(TR.synthetic_batch(t, 5, cfg, "cpu", gen)) for t in ("vqa", "nlvr", "caption")}
    fresh = {t: _cuda_batch(TR.synthetic_batch(t, 5, cfg, "cpu", gen)) for t in ("vqa", "nlvr", "caption")}    # other data, same shapes
    batches = [fixed[t] if i < 6 else fresh[t] for i, t in enumerate(order)]
    tre = TR.Trainer(m_eager, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1)
    le = [float(tre.step(b)) for b in batches]
    trg = TR.Trainer(m_graph, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1, graph=True)
    assert trg.graph
    lg = [float(trg.step(b)) for b in batches]
    assert len(trg._graphs) == 3
    for a, b in zip(lg, le):
        assert abs(a - b) <= 1e-5 * abs(b), (lg, le)
    worst = 0.0
    ref = dict(m_eager.named_parameters())
    for n, p in m_graph.named_parameters():
        if p.requires_grad:
            worst = max(worst, float((p - ref[n]).abs().max() / (ref[n].abs().max() + 1e-12)))
    assert worst <= 1e-5, worst


def test_capture_after_an_eval_forward_still_refreshes_the_weight_caches():
    """a forward between the last optimizer step and the capture (validation, a no_grad probe) leaves the
    epoch-keyed caches (K4 weight copies / pack, IO-dtype shadows) CURRENT at capture time -- without the forced new epoch in
    Trainer._capture their rebuild would not become a graph node and every replay would read weights frozen at the capture.
    Replayed training with such forwards in between must land on the eager trainer's losses and parameters."""
    import vlpet_amd.train as TR
    model, cfg = _tiny(0.0)
    m_eager, m_graph = copy.deepcopy(model).cuda(), copy.deepcopy(model).cuda()
    gen = torch.Generator().manual_seed(11)
    bs = [_cuda_batch(TR.synthetic_batch("vqa", 5, cfg, "cpu", gen)) for _ in range(6)]      # one shape: eager, capture, 4 replays
    tre = TR.Trainer(m_eager, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1)
    trg = TR.Trainer(m_graph, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1, graph=True)

    def probe(m, b):
        m.eval()
        with torch.no_grad():
            m(b["input_ids"], b["vis_inputs"], b["labels"], b["task"])
        m.train()
    le, lg = [], []
    for i, b in enumerate(bs):
        le.append(float(tre.step(b)))
        lg.append(float(trg.step(b)))
        probe(m_graph, b)                                     # between every pair of steps, incl. right before the capture step
    assert len(trg._graphs) == 1
    for a, b in zip(lg, le):
        assert abs(a - b) <= 1e-5 * abs(b), (lg, le)
    ref = dict(m_eager.named_parameters())
    worst = max(float((p - ref[n]).abs().max() / (ref[n].abs().max() + 1e-12)) for n, p in m_graph.named_parameters() if p.requires_grad)
    assert worst <= 1e-5, worst


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("lora", [False, True])
def test_deferred_finalize_is_bit_identical(graph, lora):
    """functional.DEFER_FINALIZE (round 5): the weight-gradient finalize launches of a backward queued in the library
    (vlpet_finalize_defer) and issued together at its end (vlpet_finalize_flush, 16 calls per launch) -- same per-call arithmetic, so
    losses and parameters equal the per-call launches BIT FOR BIT, eager and replayed; the queue is empty after every step."""
    import vlpet_amd.functional as VF
    import vlpet_amd.train as TR
    from vlpet_amd import _lib
    kw = dict(use_adapter=False, use_encoder_adapter_down_multihead=False, use_encoder_adapter_gating_large_x_lowrank=False,
              use_decoder_enc_attn_value_parallel_adapter_down_dim=False, unfreeze_encoder_layer_norms=False,
              use_lora=True, lora_dim=16, use_single_lora=True, lora_dropout=0.0) if lora else {}
    model, cfg = _tiny(0.0, **kw)
    gen = torch.Generator().manual_seed(9)
    bs = [_cuda_batch(TR.synthetic_batch(t, 6, cfg, "cpu", gen)) for t in ("vqa", "caption")] * 3
    lib = _lib.load()
    res = {}
    for defer in (False, True):
        VF.DEFER_FINALIZE = defer
        try:
            m = copy.deepcopy(model).cuda()
            tr = TR.Trainer(m, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1, graph=graph)
            losses = []
            for b in bs:
                losses.append(float(tr.step(b)))
                assert lib.vlpet_finalize_pending() == 0
            res[defer] = (losses, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad})
        finally:
            VF.DEFER_FINALIZE = False         # (the default: profiles/r05_deferred_finalize_ab.txt)
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for n, p in res[False][1].items():
        assert torch.equal(p, res[True][1][n]), n


def test_finalize_queue_abi():
    """vlpet_finalize_defer / _pending / _flush / _discard around two K2 backward calls: nothing is written before the flush, the flush
    (one batched launch) writes exactly what the undeferred calls write."""
    import vlpet_amd.functional as F
    from vlpet_amd import _lib
    lib = _lib.load()
    dev, dt, d, r = "cuda", torch.bfloat16, 768, 96
    g = torch.Generator(device=dev).manual_seed(2)
    st = torch.cuda.current_stream().cuda_stream
    io, tiles = F._io_dtype(torch.empty(1, dtype=dt)), F.rank_tiles(r)

    def case(M):
        x, dy = (torch.randn(M, d, device=dev, generator=g).to(dt) for _ in range(2))
        mk = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
        W = [mk(r, d), mk(r), mk(d, r), mk(d)]
        pk = F.pack_pair([W[0]], [W[1]], W[2], W[3], io, tiles)
        nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 0, io)

        def run():
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            G = [torch.full_like(w, 7.0) for w in W]
            dx = torch.empty_like(x)
            rc = lib.vlpet_parallel_adapter_bwd(dy.data_ptr(), x.data_ptr(), pk.buf.data_ptr(), dx.data_ptr(), *[t.data_ptr() for t in G],
                                                r, ws.data_ptr(), nws, M, d, tiles, 1.0, io, st)
            assert rc == 0
            return G, ws
        return run
    runs = [case(1000), case(5000)]
    plain = [run()[0] for run in runs]
    assert lib.vlpet_finalize_defer(1) == 0
    try:
        held = [run() for run in runs]
        assert lib.vlpet_finalize_defer(0) == 1
        torch.cuda.synchronize()
        assert lib.vlpet_finalize_pending() == 2
        assert all(bool((t == 7.0).all()) for G, _ in held for t in (G[0], G[2]))      # the weight gradients are still untouched
        assert lib.vlpet_finalize_flush(st) == 0 and lib.vlpet_finalize_pending() == 0
        torch.cuda.synchronize()
        for P, (G, _) in zip(plain, held):
            for a, b in zip(P, G):
                assert torch.equal(a, b)
        lib.vlpet_finalize_defer(1); runs[0](); lib.vlpet_finalize_defer(0)
        assert lib.vlpet_finalize_pending() == 1 and lib.vlpet_finalize_discard() == 0 and lib.vlpet_finalize_pending() == 0
    finally:
        lib.vlpet_finalize_defer(0); lib.vlpet_finalize_discard()


def test_graph_cache_is_bounded():
    """one captured graph per batch signature, least recently replayed evicted beyond Trainer.max_graphs."""
    import vlpet_amd.train as TR
    model, cfg = _tiny(0.0)
    m = copy.deepcopy(model).cuda()
    tr = TR.Trainer(m, cfg, lr=1e-3, total_steps=100, warmup_ratio=0.1, graph=True)
    tr.max_graphs = 2
    gen = torch.Generator().manual_seed(3)
    by_size = {n: _cuda_batch(TR.synthetic_batch("vqa", n, cfg, "cpu", gen)) for n in (3, 4, 5)}
    for n in (3, 3, 4, 4, 5, 5, 3, 3, 3):                     # third shape evicts the first; the first comes back through eager + capture
        loss = tr.step(by_size[n])
        assert bool(torch.isfinite(loss))
        assert len(tr._graphs) <= 2
    assert len(tr._graphs) == 2


def test_seed_counter_changes_the_masks_and_null_restores_them():
    """Kernel level: the K5 tail's exported mask with the same call seed at counter values 0 / 1 / 0 and without a counter."""
    from vlpet_amd import _lib
    import vlpet_amd.functional as F
    lib = _lib.load()
    M, d = 300, 768
    g = torch.Generator().manual_seed(1)
    y, x1 = (torch.randn(M, d, generator=g).cuda().to(torch.bfloat16) for _ in range(2))
    out = torch.empty_like(y)
    st = torch.cuda.current_stream().cuda_stream

    def mask(p=0.1, seed=0x1234567):
        k = torch.empty(M, d, dtype=torch.uint8, device="cuda")
        rc = lib.vlpet_sublayer_tail_fwd(y.data_ptr(), x1.data_ptr(), None, None, out.data_ptr(), None, None, None, k.data_ptr(),
                                         M, d, 1e-5, p, seed, 0, F._io_dtype(y), st)
        assert rc == 0
        torch.cuda.synchronize()
        return k.cpu()
    base = mask()
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    assert lib.vlpet_set_seed_counter(ctr.data_ptr()) == 0
    m0 = mask()
    ctr += 1
    m1 = mask()
    ctr -= 1
    m0b = mask()
    lib.vlpet_set_seed_counter(None)
    again = mask()
    assert torch.equal(base, m0) and torch.equal(m0, m0b) and torch.equal(base, again)
    assert not torch.equal(m0, m1)
    assert 0.88 < float(m1.float().mean()) < 0.92
    assert abs(float((m0 == m1).float().mean()) - (0.81 + 0.01)) < 0.02      # independent masks agree on 0.9^2 + 0.1^2 of the elements


def test_replayed_training_with_dropout_reduces_the_loss():
    """The whole stochastic path under replay: the K5 / FFN / attention dropout kernels take fresh masks each step through the
    step counter, the backward regenerates them from the same value.  Had every replay repeated one mask the fit would be the
    (much faster) overfit of a fixed sub-network; a forward / backward mask mismatch would not converge at all."""
    import vlpet_amd.train as TR
    model, cfg = _tiny(0.1)
    model.cuda()
    tr = TR.Trainer(model, cfg, lr=5e-3, total_steps=80, warmup_ratio=0.05, graph=True)
    b = _cuda_batch(TR.synthetic_batch("caption", 8, cfg, "cpu", torch.Generator().manual_seed(2)))
    losses = [float(tr.step(b)) for _ in range(40)]
    assert len(tr._graphs) == 1 and int(tr.seed_ctr.item()) == 40
    assert all(l == l for l in losses)
    assert sum(losses[-5:]) / 5 < sum(losses[:3]) / 3 - 0.1, (losses[:3], losses[-5:])
    # the replayed steps did not all see the same masks: the loss of a fixed batch under a fixed mask falls monotonically at this
    # step size, with fresh masks it jitters
    d = [losses[i + 1] - losses[i] for i in range(10, 39)]
    assert any(x > 0 for x in d)


_NLVR_REPLAY_CHILD = r"""
import copy, sys, torch
sys.path.insert(0, {tests!r}); sys.path.insert(0, {root!r})
import test_gpu_graph as T
import vlpet_amd.train as TR
model, cfg = T._tiny(0.0)
m_eager, m_graph = copy.deepcopy(model).cuda(), copy.deepcopy(model).cuda()
gen = torch.Generator().manual_seed(7)
tasks = ("gqa", "nlvr", "caption")          # NLVR captured AFTER one shape and BEFORE another: the order that showed the fault
batches = {{t: T._cuda_batch(TR.synthetic_batch(t, 48, cfg, "cpu", gen)) for t in tasks}}     # 48 x 72 image-order ids > 3,072: torch's sort-based embedding backward
order = tasks * 5
tre = TR.Trainer(m_eager, cfg, lr=1e-3, total_steps=40, warmup_ratio=0.1)
le = [float(tre.step(batches[t])) for t in order]
trg = TR.Trainer(m_graph, cfg, lr=1e-3, total_steps=40, warmup_ratio=0.1, graph=True)
lg = []
for t in order:
    lg.append(float(trg.step(batches[t])))
    torch.cuda.synchronize()
assert len(trg._graphs) == 3
for a, b in zip(lg, le):
    assert abs(a - b) <= 1e-4 * abs(b), (lg, le)
print("ok")
"""


def test_replayed_nlvr_step_survives_later_captures():
    """Round 6: the replayed full-batch BART / LoRA bench died with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in the SECOND replay of the NLVR
    step whenever another shape had been captured before AND after it.  Cause: NLVR hands the visual embedding one image-order id per
    visual token (B x 72 > 3,072 indices), torch's embedding backward then takes its sort-based path, and that kernel sequence does not
    survive hipGraph replay on this stack; visual._order_lookup runs the few-row table as a one-hot GEMM instead.  Child process (a memory
    fault aborts the interpreter): three shapes with NLVR in the middle, five rounds, replayed losses equal the eager trainer's."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _NLVR_REPLAY_CHILD.format(tests=os.path.dirname(os.path.abspath(__file__)), root=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_batches_written_into_the_captured_input_buffers_replay_without_a_copy():
    """Trainer.input_buffers: a loader may write each new batch INTO the captured step's input tensors and pass those to step(); the
    losses equal the eager trainer's on the same data, and step() leaves the buffers' addresses alone (nothing is copied)."""
    import vlpet_amd.train as TR
    model, cfg = _tiny(0.0)
    m_eager, m_graph = copy.deepcopy(model).cuda(), copy.deepcopy(model).cuda()
    gen = torch.Generator().manual_seed(9)
    data = [_cuda_batch(TR.synthetic_batch("vqa", 5, cfg, "cpu", gen)) for _ in range(5)]
    tre = TR.Trainer(m_eager, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1)
    le = [float(tre.step(b)) for b in data]
    trg = TR.Trainer(m_graph, cfg, lr=1e-2, total_steps=20, warmup_ratio=0.1, graph=True)
    assert trg.input_buffers(data[0]) is None                     # nothing captured yet
    lg = [float(trg.step(data[0])), float(trg.step(data[1]))]     # eager, then capture + replay
    buf = trg.input_buffers(data[2])
    assert buf is not None and buf["task"] == "vqa"
    ptrs = [t.data_ptr() for _, _, t in TR.Trainer._leaves(buf)]
    for b in data[2:]:
        for (_, _, src), (_, _, dst) in zip(TR.Trainer._leaves(b), TR.Trainer._leaves(buf)):
            dst.copy_(src)                                        # the "loader" writes in place
        lg.append(float(trg.step(buf)))
    assert [t.data_ptr() for _, _, t in TR.Trainer._leaves(trg.input_buffers(buf))] == ptrs
    for a, b in zip(lg, le):
        assert abs(a - b) <= 1e-5 * abs(b), (lg, le)
