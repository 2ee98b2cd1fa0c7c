"""GPU: K4 visual projection (HIP GEMM + LayerNorm kernel, MFMA weight gradient) -- the drop-in
VisualEmbedding module against the golden fixtures of the reference class and against the oracle at the
real shape (feat_dim 2048 -> 768)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from gpu_cases import rel_err  # noqa: E402
from oracle import vlpet_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k]) for k in z.files}


def build(d, feat_dim, rms, vocab=200):
    import vlpet_amd.host.bart as HB
    from vlpet_amd.visual import VisualEmbedding
    cfg = HB.vlpet_config(d_model=d, feat_dim=feat_dim)
    table = torch.nn.Embedding(vocab, d)
    return VisualEmbedding(cfg, table, rms_norm=rms), table


@pytest.mark.parametrize("name", ["k4_bart_d64_f128", "k4_bart_nlvr_d64_f128", "k4_bart_d128_f256", "k4_t5_d64_f128"])
def test_k4_golden(name):
    g = load(name)
    d, F, B, N = [int(v) for v in g["meta"]]
    rms = bool(int(g["rms"]))
    ve, table = build(d, F, rms)
    with torch.no_grad():
        fe, pe = ve.feat_embedding, ve.absolute_vis_pos_embedding
        fe[0].weight.copy_(g["feat_w"]); fe[0].bias.copy_(g["feat_b"]); fe[1].weight.copy_(g["feat_ln_w"])
        pe[0].weight.copy_(g["pos_w"]); pe[0].bias.copy_(g["pos_b"]); pe[1].weight.copy_(g["pos_ln_w"])
        if not rms:
            fe[1].bias.copy_(g["feat_ln_b"]); pe[1].bias.copy_(g["pos_ln_b"])
        else:
            fe[1].variance_epsilon = float(g["eps"]); pe[1].variance_epsilon = float(g["eps"])
        ve.img_order_embedding.weight.copy_(g["img_table"]); table.weight.copy_(g["obj_table"])
    ve = ve.cuda()
    img_ids = torch.from_numpy(g["img_ids"]).cuda() if g["img_ids"].size else None
    obj_ids = torch.from_numpy(g["obj_ids"]).cuda() if g["obj_ids"].size else None
    out = ve(g["feats"].cuda(), g["pos"].cuda(), img_ids, obj_ids)
    assert rel_err(out, g["out"]) <= 1e-3
    out.backward(g["dy"].cuda())
    fe, pe = ve.feat_embedding, ve.absolute_vis_pos_embedding
    checks = [(fe[0].weight.grad, "d_feat_w"), (fe[0].bias.grad, "d_feat_b"), (fe[1].weight.grad, "d_feat_ln_w"),
              (pe[0].weight.grad, "d_pos_w"), (pe[0].bias.grad, "d_pos_b"), (pe[1].weight.grad, "d_pos_ln_w"),
              (ve.img_order_embedding.weight.grad, "d_img_table"), (ve.obj_order_embedding.weight.grad, "d_obj_table")]
    if not rms:
        checks += [(fe[1].bias.grad, "d_feat_ln_b"), (pe[1].bias.grad, "d_pos_ln_b")]
    for got, key in checks:
        assert rel_err(got, g[key]) <= 1e-3, key


FORMS = ["gemm", "library", "fused"]      # visproj.K4_FORM: the round-5 tiled GEMM (default), library GEMM + norm pass, the round-2 fused kernel


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-2)])
def test_k4_real_shape_vs_oracle(dtype, tol, form, monkeypatch):
    import vlpet_amd.visproj as VP
    monkeypatch.setattr(VP, "K4_FORM", form)                   # (fp32 IO runs the fused kernel whatever the form)
    torch.manual_seed(3)
    B, N, F, d = 9, 36, 2048, 768            # 324 rows: a partial last workgroup
    ve, table = build(d, F, False, vocab=300)
    with torch.no_grad():
        for p in ve.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    feats = torch.randn(B, N, F).to(dtype)
    pos = torch.rand(B, N, 4)
    dy = torch.randn(B, N, d).to(dtype)
    fe, pe = ve.feat_embedding, ve.absolute_vis_pos_embedding
    names = [fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias,
             ve.img_order_embedding.weight, table.weight]
    ref = [t.detach().clone().requires_grad_(True) for t in names]
    out_ref = O.visual_embedding(feats.float(), pos, *ref[:8], ref[8], ref[9])
    out_ref.backward(dy.float())
    ve = ve.cuda()
    out = ve(feats.cuda(), pos.cuda())
    assert rel_err(out, out_ref) <= tol
    out.backward(dy.cuda())
    got = [fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias,
           ve.img_order_embedding.weight]
    for a, b in zip(got, ref[:9]):
        assert rel_err(a.grad, b.grad) <= tol


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("B", [500, 833])
def test_k4_forward_at_the_bench_rows_vs_oracle(B, form, monkeypatch):
    """The K4 forward at BASELINE configs[1]'s full size (vqa: 500 x 36 = 18,000 rows; gqa: 833 x 36 = 29,988 rows; feat_dim 2048 ->
    768, bf16) against oracle.visual_embedding (src/modeling_bart.py:157, 162-190) -- the fixtures and the 324-row oracle case run
    three workgroups, the bench 146+ (VERDICT r03 missing #5).  All three forms; the default ("gemm") runs 74 / 85 teams of three
    workgroups here, the second in several passes over the row blocks."""
    import vlpet_amd.visproj as VP
    monkeypatch.setattr(VP, "K4_FORM", form)
    torch.manual_seed(11)
    N, F, d = 36, 2048, 768
    ve, table = build(d, F, False, vocab=300)
    with torch.no_grad():
        for p in ve.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    feats = torch.randn(B, N, F).to(torch.bfloat16)
    pos = torch.rand(B, N, 4)
    fe, pe = ve.feat_embedding, ve.absolute_vis_pos_embedding
    names = [fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias,
             ve.img_order_embedding.weight, table.weight]
    with torch.no_grad():
        out_ref = O.visual_embedding(feats.float(), pos, *[t.detach() for t in names[:8]], names[8].detach(), names[9].detach())
    ve = ve.cuda()
    with torch.no_grad():
        out = ve(feats.cuda(), pos.cuda())
    assert out.shape == (B, N, d) and rel_err(out, out_ref) <= 1e-2
    if form == "gemm":
        assert VP._VisProjFn.last_status is not None and int(VP._VisProjFn.last_status.view(torch.int32)[0].item()) == 0


def _gemm_abi(M, F, d, rms, form, bm, with_r=True, seed=0, mean_shift=0.0, ws_keep=None, hold=0):
    """vlpet_visproj_fwd_gemm_cfg directly; returns (out, xhat, rstd, mean, status) and the fp32 reference tensors.  hold = N: N
    workgroups of vlpet_test_hold_cus (100 KiB of LDS each: one per CU, and no K4 workgroup fits beside one) occupy CUs from a second
    stream while the call runs, and are released once it has finished."""
    import time
    from vlpet_amd import _lib
    lib = _lib.load()
    flag = side = None
    if hold:
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        side = torch.cuda.Stream(priority=-1)     # (a high-priority stream has hardware queues of its own: a normal one may share the default stream's and serialise with it)
        torch.cuda.synchronize()
        assert lib.vlpet_test_hold_cus(hold, 100 * 1024, flag.data_ptr(), 6000, side.cuda_stream) == 0
        time.sleep(0.05)        # (the holders are resident before the call under test is issued)
    g = torch.Generator(device="cuda").manual_seed(seed)
    feats = torch.randn(M, F, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(d, F, device="cuda", generator=g) * (1.0 / F ** 0.5)).to(torch.bfloat16)
    b = torch.randn(d, device="cuda", generator=g) * 0.3 + mean_shift
    gam = 1.0 + 0.2 * torch.randn(d, device="cuda", generator=g)
    bet = None if rms else 0.1 * torch.randn(d, device="cuda", generator=g)
    R = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16) if with_r else None
    out = torch.full((M, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    xhat = torch.full((M, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    rstd = torch.full((M,), float("nan"), device="cuda"); mean = torch.full((M,), float("nan"), device="cuda")
    nws = lib.vlpet_visproj_gemm_workspace_bytes(M, F, d)
    assert nws > 0
    ws = ws_keep if ws_keep is not None and ws_keep.numel() >= nws else torch.zeros(nws, dtype=torch.uint8, device="cuda")
    ptr = lambda t: None if t is None else t.data_ptr()
    eps = 1e-6 if rms else 1e-5
    rc = lib.vlpet_visproj_fwd_gemm_cfg(feats.data_ptr(), w.data_ptr(), b.data_ptr(), gam.data_ptr(), ptr(bet), ptr(R), out.data_ptr(),
                                        xhat.data_ptr(), rstd.data_ptr(), mean.data_ptr(), ws.data_ptr(), nws, M, F, d, eps, int(rms),
                                        _lib.VLPET_BF16, form, bm, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.current_stream().synchronize()       # (not the device: the holders are still spinning)
    if hold:
        t_held = time.time()
        flag.fill_(1)
    torch.cuda.synchronize()
    if hold:
        assert time.time() - t_held < 3.0, "the holders were not released by the flag (they ran into their own time bound)"
    pre = feats.float() @ w.float().t() + b
    if rms:
        r_rstd = torch.rsqrt(pre.pow(2).mean(-1) + eps); r_mean = torch.zeros(M, device="cuda")
    else:
        r_mean = pre.mean(-1); r_rstd = torch.rsqrt(pre.var(-1, unbiased=False) + eps)
    r_xhat = (pre - r_mean[:, None]) * r_rstd[:, None]
    r_out = r_xhat * gam + (bet if bet is not None else 0.0) + (R.float() if R is not None else 0.0)
    status = int(ws[:4].view(torch.int32)[0].item())
    xb = lib.vlpet_visproj_gemm_exchange_bytes(d)
    assert int(ws[256:256 + xb].count_nonzero().item()) == 0    # every consumer cleared what it read (or the repair pass did): the exchange area is clean again
    hdr = ws[:16].view(torch.int32).tolist()
    assert hdr[1] == hdr[2] and hdr[3] == 0                      # every tile that gave up has been repaired
    assert int(ws[256 + xb:256 + xb + 65536].count_nonzero().item()) == 0     # ... and its flag cleared (the fixed 64-KiB flag area)
    return (out, xhat, rstd, mean, status), (r_out, r_xhat, r_rstd, r_mean)


@pytest.mark.parametrize("form,bm", [(f, b) for f in (1, 2, 3, 4, 5, 6) for b in (128, 256)] + [(1, 192), (4, 192), (0, 0)])
@pytest.mark.parametrize("M,F,d,rms", [(1, 64, 256, False), (300, 512, 768, False), (1000, 2048, 768, True), (12000, 512, 512, False),
                                       (18700, 2048, 768, False), (29988, 2048, 768, False), (23000, 512, 1024, True)])
def test_k4_gemm_kernel_through_the_abi(M, F, d, rms, form, bm):
    """csrc/visproj_gemm.hip through the C ABI, every ring form and both tile heights: one row, ragged last row block, one to four
    column tiles per team, several passes of the teams over the row blocks (29,988 rows: 118 / 235 row blocks for 85 teams), the rms
    form -- out, xhat, rstd (and mean) against an fp32 reference of the same bf16 inputs; the status word must stay 0."""
    got, ref = _gemm_abi(M, F, d, rms, form, bm)
    assert got[4] == 0
    assert rel_err(got[0], ref[0]) <= 1e-2 and rel_err(got[1], ref[1]) <= 1e-2
    assert rel_err(got[2], ref[2]) <= 1e-4
    if not rms:
        assert float((got[3] - ref[3]).abs().max()) <= 1e-4 * float(ref[3].abs().max() + 1.0)


def test_k4_gemm_statistics_survive_a_large_common_offset():
    """Rows whose mean is 50x their spread (a bias-dominated projection): the per-tile statistics are combined with Chan's formula, not
    as E[x^2] - mean^2 -- rstd must still match to fp32 rounding of the sums."""
    got, ref = _gemm_abi(2000, 512, 768, False, 0, 0, with_r=False, seed=4, mean_shift=40.0)
    assert got[4] == 0
    assert rel_err(got[2], ref[2]) <= 2e-3 and rel_err(got[1], ref[1]) <= 1e-2


_ONE_POLL = 31 << 8          # form bits 8-12 = 31: "log2 of the bound" 31 is read as the special value ONE poll (csrc/visproj_gemm.hip)
_GIVE_UP = 30 << 8           # ... 30: every workgroup takes the give-up path without polling


def _gemm_check(got, ref, rms):
    assert rel_err(got[0], ref[0]) <= 1e-2 and rel_err(got[1], ref[1]) <= 1e-2
    assert rel_err(got[2], ref[2]) <= 1e-4
    if not rms:
        assert float((got[3] - ref[3]).abs().max()) <= 1e-4 * float(ref[3].abs().max() + 1.0)


@pytest.mark.parametrize("M,F,d,rms", [(18700, 2048, 768, False), (10800, 2048, 768, True)])
def test_k4_gemm_with_64_cus_held_by_another_stream(M, F, d, rms):
    """VERDICT r05 weak #1 / ADVICE: the statistics exchange presumes that a team's workgroups run at the same time.  Here 64 CUs (8 per
    XCD) are held by a spinning kernel on a second stream for the whole call: fewer CUs than workgroups, so part of the grid waits
    for others to finish.  The output must still match the reference (whether or not a workgroup gave up on the way -- a give-up is
    repaired inside the call), and the workspace must be left clean."""
    got, ref = _gemm_abi(M, F, d, rms, 0, 0, seed=11, hold=64)
    _gemm_check(got, ref, rms)


@pytest.mark.parametrize("mode", ["all", "one_poll"])
@pytest.mark.parametrize("M,F,d,rms", [(2000, 512, 768, False), (1500, 512, 768, True), (900, 512, 1024, False), (18700, 2048, 768, False)])
def test_k4_gemm_workgroups_that_give_up_are_repaired_inside_the_call(M, F, d, rms, mode):
    """The give-up path itself.  "all": every workgroup gives up on its partners without polling, so EVERY tile leaves its pre-norm rows
    and is normalised by the repair kernel; "one_poll": the polling bound is one poll, so a workgroup gives up exactly when a partner has
    not published by then (most of them at the large size, possibly none at the small ones -- whatever the timing gives).  The call must
    (i) return rows that match the reference all the same -- the repair kernel re-normalises the flagged tiles from the complete
    per-tile statistics --, (ii) report it in the status word, (iii) leave the exchange area clean although late producers wrote granules
    after their consumer had given up (checked inside _gemm_abi); a plain call on the SAME workspace afterwards is right and does not
    touch the counters.  (Holding all but a few CUs from a second stream does not work as a trigger: while a high-priority kernel still
    has workgroups to place, no other kernel is dispatched at all -- profiles/r06_in_launch_reduce_ab.txt (4).)"""
    from vlpet_amd import _lib
    ws = torch.zeros(_lib.load().vlpet_visproj_gemm_workspace_bytes(M, F, d), dtype=torch.uint8, device="cuda")
    got, ref = _gemm_abi(M, F, d, rms, _GIVE_UP if mode == "all" else _ONE_POLL, 128, seed=12, ws_keep=ws)
    _gemm_check(got, ref, rms)
    hdr = ws[:16].view(torch.int32).tolist()
    if mode == "all":
        tiles = ((M + 127) // 128) * (d // 256)
        assert got[4] != 0 and hdr[1] == tiles, (hdr, tiles)
    given_up = hdr[1]
    again, ref2 = _gemm_abi(M, F, d, rms, 0, 128, seed=13, ws_keep=ws)
    _gemm_check(again, ref2, rms)
    assert ws[:16].view(torch.int32).tolist()[1] == given_up


def test_k4_gemm_repeated_launches_share_one_exchange_area():
    """The exchange area is zeroed once and left clean by every launch: 20 back-to-back launches at two sizes and both tile heights on
    ONE area (a side stream) give identical results and leave it all-zero (checked inside _gemm_abi)."""
    from vlpet_amd import _lib
    ws = torch.zeros(_lib.load().vlpet_visproj_gemm_workspace_bytes(29988, 512, 768), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        base = _gemm_abi(29988, 512, 768, False, 0, 0, seed=9, ws_keep=ws)[0]
        for i in range(10):
            again = _gemm_abi(29988, 512, 768, False, 0, 256 if i % 2 else 128, seed=9, ws_keep=ws)[0]
            assert again[4] == 0 and torch.equal(again[2], base[2]) and rel_err(again[0], base[0].float()) <= 1e-2
            small = _gemm_abi(700, 512, 768, False, 0, 0, seed=9, ws_keep=ws)[0]
            assert small[4] == 0


@pytest.mark.parametrize("M,F,d", [(1, 256, 384), (31, 256, 384), (33, 512, 768), (4097, 2048, 768), (18700, 2048, 768)])
def test_k4_weight_gradient_tiled_gemm(M, F, d):
    """vlpet_visproj_wgrad in its round-3 form (csrc/visproj_wgrad.hip: [384 x 256] tiles, split-K over row chunks) against
    fp32 matmuls of the same bf16 inputs: dW = dpre^T feats, db = column sums of dpre; one row, ragged last steps, one tile."""
    from vlpet_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    dpre = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
    feats = torch.randn(M, F, device="cuda", generator=g).to(torch.bfloat16)
    dw = torch.full((d, F), float("nan"), device="cuda"); db = torch.full((d,), float("nan"), device="cuda")
    nws = lib.vlpet_visproj_wgrad_workspace_bytes(M, F, d)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    rc = lib.vlpet_visproj_wgrad(dpre.data_ptr(), feats.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, M, F, d,
                                 _lib.VLPET_BF16, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref_w = dpre.float().t() @ feats.float()
    ref_b = dpre.float().sum(0)
    assert rel_err(dw, ref_w) <= 1e-4 and rel_err(db, ref_b) <= 1e-4      # exact products of bf16 inputs, fp32 sums


# ------------------------------------------------------------------------------------------------ position / order branch (csrc/vispos.hip)
def _pos_oracle(pos, w, b, g, be, img_t, obj_t, img_ids, obj_ids, rms, eps):
    """oracle.visual_embedding with a zero feature branch = the position branch + the order embeddings alone (src/modeling_bart.py:162-183)"""
    B, N, _ = pos.shape
    d = w.shape[0]
    feats = torch.zeros(B, N, 8)
    return O.visual_embedding(feats, pos, torch.zeros(d, 8), torch.zeros(d), None, None, w, b, g, be, img_t, obj_t,
                              img_order_ids=img_ids, obj_order_ids=obj_ids, eps=eps, rms=rms)


@pytest.mark.parametrize("d,B,N,ids,tab_dt,rms,io", [
    (768, 9, 36, "default", "f32", False, "bf16"),
    (768, 9, 36, "default", "f32", False, "f32"),
    (768, 7, 72, "per_sample", "f32", False, "bf16"),          # NLVR: explicit image ids per sample, two images
    (768, 5, 36, "broadcast", "bf16", False, "bf16"),          # a frozen bf16 token table, [1, N] ids
    (768, 6, 36, "default", "f32", True, "bf16"),              # T5LayerNorm
    (256, 3, 10, "per_sample", "f32", False, "f32"),
    (1024, 4, 36, "per_sample", "bf16", True, "bf16"),
    (768, 500, 36, "default", "f32", False, "bf16"),           # configs[1] vqa rows (18,000): every workgroup several rows per wave
    (768, 4, 36, "none", "f32", False, "bf16"),                # use_vis_order_embedding off
])
def test_position_branch_kernel_through_the_abi_vs_oracle(d, B, N, ids, tab_dt, rms, io):
    from vlpet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(5)
    M, n_img, V = B * N, (4 if d == 1024 else 2), 97
    eps = 1e-6 if rms else 1e-5
    pos = torch.rand(B, N, 4)
    pos[..., 1] += pos[..., 0]; pos[..., 3] += pos[..., 2]
    w, b = torch.randn(d, 5) * 0.3, torch.randn(d) * 0.1
    g, be = 1 + 0.1 * torch.randn(d), (None if rms else 0.1 * torch.randn(d))
    tdt = torch.bfloat16 if tab_dt == "bf16" else torch.float32
    img_t, obj_t = (torch.randn(n_img, d) * 0.5).to(tdt), (torch.randn(V, d) * 0.5).to(tdt)
    img_ids = obj_ids = None
    if ids == "per_sample":
        img_ids, obj_ids = torch.randint(0, n_img, (B, N)), torch.randint(0, V, (B, N))
    elif ids == "broadcast":
        img_ids, obj_ids = torch.randint(0, n_img, (1, N)), torch.randint(0, N, (1, N))
    tabs = ids != "none"
    leaves = [t.clone().requires_grad_(True) for t in (w, b, g)] + ([be.clone().requires_grad_(True)] if be is not None else [None])
    img_leaf = img_t.float().clone().requires_grad_(True) if tabs else None
    ref = _pos_oracle(pos, leaves[0], leaves[1], leaves[2], leaves[3], img_leaf, obj_t.float() if tabs else None, img_ids, obj_ids, rms, eps)
    iot = torch.float32 if io == "f32" else torch.bfloat16
    dout = torch.randn(B, N, d).to(iot)
    ref.backward(dout.float())
    dev = "cuda"
    c = lambda t: None if t is None else t.to(dev).contiguous()
    pos_c, w_c, b_c, g_c, be_c, img_c, obj_c = c(pos), c(w), c(b), c(g), c(be), c(img_t) if tabs else None, c(obj_t) if tabs else None
    ii, oi = c(img_ids), c(obj_ids)
    out = torch.empty(B, N, d, dtype=iot, device=dev)
    iod = _lib.VLPET_F32 if io == "f32" else _lib.VLPET_BF16
    tdd = _lib.VLPET_BF16 if tab_dt == "bf16" else _lib.VLPET_F32
    P = lambda t: None if t is None else t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    bstride = lambda t: 0 if (t is None or t.shape[0] == 1) else N
    assert lib.vlpet_vispos_applies(d, n_img if tabs else 0) == 1
    rc = lib.vlpet_vispos_fwd(P(pos_c), P(w_c), P(b_c), P(g_c), P(be_c), P(img_c), tdd, n_img if tabs else 0, P(ii), bstride(ii),
                              P(obj_c), tdd, V if tabs else 0, P(oi), bstride(oi), P(out), M, N, d, eps, int(rms), iod, st)
    assert rc == 0
    torch.cuda.synchronize()
    # the kernel rounds the fp32 value once to the IO dtype: one bf16 ulp of the largest entries
    tol_el = 2e-5 if io == "f32" else 2.0 ** -8
    err = (out.float().cpu() - ref.detach()).abs().max().item()
    assert err <= tol_el * max(1.0, ref.detach().abs().max().item()), err
    dw, db, dg = torch.full((d, 5), 7.0, device=dev), torch.full((d,), 7.0, device=dev), torch.full((d,), 7.0, device=dev)
    dbe = None if rms else torch.full((d,), 7.0, device=dev)
    dimg = torch.full((n_img, d), 7.0, device=dev) if tabs else None
    nws = lib.vlpet_vispos_bwd_workspace_bytes(M, d, n_img if tabs else 0)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    dout_c = c(dout)
    rc = lib.vlpet_vispos_bwd(P(dout_c), P(pos_c), P(w_c), P(b_c), P(g_c), n_img if tabs else 0, P(ii), bstride(ii),
                              P(dw), P(db), P(dg), P(dbe), P(dimg), P(ws), nws, M, N, d, eps, int(rms), iod, st)
    assert rc == 0
    torch.cuda.synchronize()
    pairs = [(dw, leaves[0].grad, "dw"), (db, leaves[1].grad, "db"), (dg, leaves[2].grad, "dgamma")]
    if not rms:
        pairs.append((dbe, leaves[3].grad, "dbeta"))
    if tabs:
        pairs.append((dimg, img_leaf.grad, "dimg"))
    for got, want, name in pairs:
        assert rel_err(got, want) <= 2e-4, (name, rel_err(got, want))
    # deterministic: a second call writes the same bits
    dw2 = torch.empty_like(dw)
    rc = lib.vlpet_vispos_bwd(P(dout_c), P(pos_c), P(w_c), P(b_c), P(g_c), n_img if tabs else 0, P(ii), bstride(ii),
                              P(dw2), P(db), P(dg), P(dbe), P(dimg), P(ws), nws, M, N, d, eps, int(rms), iod, st)
    assert rc == 0 and torch.equal(dw, dw2)


def test_position_branch_abi_rejects_what_it_does_not_cover():
    from vlpet_amd import _lib
    lib = _lib.load()
    assert lib.vlpet_vispos_applies(64, 2) == 0 and lib.vlpet_vispos_applies(768, 5) == 0 and lib.vlpet_vispos_applies(1280, 2) == 0
    t = torch.zeros(768 * 5, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vlpet_vispos_fwd(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 0, 0, None, 0, None, 0, 0, None, 0,
                                t.data_ptr(), 10, 3, 768, 1e-5, 0, _lib.VLPET_BF16, st) == -1      # VLPET_E_SHAPE: M not a multiple of N
    assert lib.vlpet_vispos_fwd(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, None, 0, 0, None, 0, None, 0, 0, None, 0,
                                t.data_ptr(), 9, 3, 768, 1e-5, 0, _lib.VLPET_BF16, st) == -5       # VLPET_E_NULL: LayerNorm without beta


@pytest.mark.parametrize("rms", [False, True])
@pytest.mark.parametrize("nlvr", [False, True])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-2)])
def test_k4_module_with_a_frozen_token_table_takes_the_position_kernel(dtype, tol, nlvr, rms, monkeypatch):
    """The launch scripts' case: the shared token table (= obj_order_embedding) frozen, everything else of the visual embedding trainable
    (trainer_base.py:308-542).  Output and every gradient against the oracle, and the position branch must have run as the HIP kernel."""
    import vlpet_amd.visproj as VP
    calls = []
    orig = VP._VisPosFn.apply
    monkeypatch.setattr(VP._VisPosFn, "apply", lambda *a: (calls.append(1), orig(*a))[1])
    torch.manual_seed(4)
    B, N, F, d = 6, (72 if nlvr else 36), 2048, 768
    ve, table = build(d, F, rms, vocab=300)          # rms: the T5 variant (T5LayerNorm on both branches, src/modeling_t5.py:56-73)
    with torch.no_grad():
        for p in ve.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    table.weight.requires_grad_(False)
    feats = torch.randn(B, N, F).to(dtype)
    pos = torch.rand(B, N, 4)
    dy = torch.randn(B, N, d).to(dtype)
    img_ids = obj_ids = None
    if nlvr:        # src/nlvr_model.py:162-170: two images, 36 objects each
        img_ids = torch.cat([torch.zeros(B, 36, dtype=torch.long), torch.ones(B, 36, dtype=torch.long)], 1)
        obj_ids = torch.arange(36).repeat(2).unsqueeze(0).expand(B, -1).contiguous()
    fe, pe = ve.feat_embedding, ve.absolute_vis_pos_embedding
    if rms:
        names = [fe[0].weight, fe[0].bias, fe[1].weight, pe[0].weight, pe[0].bias, pe[1].weight, ve.img_order_embedding.weight]
        ref = [t.detach().clone().requires_grad_(True) for t in names]
        out_ref = O.visual_embedding(feats.float(), pos, ref[0], ref[1], ref[2], None, ref[3], ref[4], ref[5], None, ref[6], table.weight.detach(),
                                     img_order_ids=img_ids, obj_order_ids=obj_ids, eps=fe[1].variance_epsilon, rms=True)
    else:
        names = [fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, pe[0].weight, pe[0].bias, pe[1].weight, pe[1].bias, ve.img_order_embedding.weight]
        ref = [t.detach().clone().requires_grad_(True) for t in names]
        out_ref = O.visual_embedding(feats.float(), pos, *ref[:8], ref[8], table.weight.detach(), img_order_ids=img_ids, obj_order_ids=obj_ids)
    out_ref.backward(dy.float())
    ve = ve.cuda()
    out = ve(feats.cuda(), pos.cuda(), None if img_ids is None else img_ids.cuda(), None if obj_ids is None else obj_ids.cuda())
    assert calls, "the position branch did not take csrc/vispos.hip"
    assert rel_err(out, out_ref) <= tol
    out.backward(dy.cuda())
    for a, b in zip(names, ref):
        assert rel_err(a.grad, b.grad) <= tol
