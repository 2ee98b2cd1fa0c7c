"""Shared case runners for the GPU parity tests and tools/gpu_diag.py: run the HIP path through
vlpet_amd.functional and the CPU oracle on the same seeded inputs, return per-tensor errors."""
import math

import numpy as np
import torch

from oracle import vlpet_oracle as O


def rel_err(a: torch.Tensor, ref: torch.Tensor) -> float:
    a = a.detach().float().cpu()
    ref = ref.detach().float().cpu()
    if not torch.isfinite(a).all():
        return float("inf")
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-6))


def col_rel_err(a: torch.Tensor, ref: torch.Tensor) -> float:
    """Worst per-element relative error of a vector (bias gradients): |a_j - ref_j| / max(|ref_j|, 0.5 * mean|ref|).
    rel_err above is a global norm and hides small-magnitude columns; this one does not."""
    a = a.detach().float().cpu().reshape(-1)
    ref = ref.detach().float().cpu().reshape(-1)
    if not torch.isfinite(a).all():
        return float("inf")
    floor = 0.5 * float(ref.abs().mean()) + 1e-12
    return float(((a - ref).abs() / ref.abs().clamp_min(floor)).max())


EL_FLOOR = 0.02         # default floor of the element-wise metric as a fraction of max|ref|
K1_EL_FLOOR = 0.06      # ... for the outputs and gradients of the K1 op (run_k1): argued in tests/test_gpu_cols.py, with its 0.1 bound


def el_rel_err(a: torch.Tensor, ref: torch.Tensor, floor: float = EL_FLOOR) -> float:
    """Element-wise |a - ref| / (|ref| + floor * max|ref|), worst element: every entry is held to its OWN magnitude, small ones
    against a floor of `floor` of the tensor's maximum (2 % unless a caller argues another: ADVICE r05).  rel_err above is a global norm
    (max-abs over max-abs-ref) and cannot see an entry that is wrong by its own size while small next to the largest one."""
    a = a.detach().float().cpu()
    ref = ref.detach().float().cpu()
    if not torch.isfinite(a).all():
        return float("inf")
    return float(((a - ref).abs() / (ref.abs() + floor * ref.abs().max().clamp_min(1e-12))).max())


def _rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def make_k1(seed, M, d, r, rg, nh, wscale=None):
    g = torch.Generator().manual_seed(seed)
    ws = wscale if wscale is not None else 1.0 / math.sqrt(d)
    wu_s = wscale if wscale is not None else 1.0 / math.sqrt(max(r, 1))
    t = dict(x1=_rand(g, M, d), x2=_rand(g, M, d), dy=_rand(g, M, d),
             wd=_rand(g, r, d, scale=ws), bd=_rand(g, r, scale=0.1), wu=_rand(g, d, r, scale=wu_s),
             bu=_rand(g, d, scale=0.1), wgd=_rand(g, rg, d, scale=ws), bgd=_rand(g, rg, scale=0.1),
             wgu=_rand(g, d, rg, scale=wu_s), bgu=_rand(g, d, scale=0.1))
    return t


def run_k1(dtype, M=224, d=768, r=96, rg=96, nh=4, gate_mode=1, delta_scale=1.0, x2_scale=1.0, gate_scale=1.0,
           seed=0, tensors=None, col_errs=None, el_errs=None):
    """returns dict name -> relative error (max-abs / max-abs-ref) for y, dx1, dx2 and the 8 grads; ``col_errs`` (a dict)
    additionally receives the per-column relative errors of the four bias gradients"""
    import vlpet_amd.functional as F
    t = tensors if tensors is not None else make_k1(seed, M, d, r, rg, nh)
    dev = "cuda"
    act = {k: t[k].to(dtype) for k in ("x1", "x2", "dy")}
    # oracle sees the same (rounded) activations, fp32 weights, fp32 arithmetic
    has_gate = gate_mode != 0
    y_ref, g_ref = O.k1_fwd_bwd(act["x1"].float(), act["x2"].float(), t["wd"], t["bd"], t["wu"], t["bu"],
                                t.get("wgd"), t.get("bgd"), t.get("wgu"), t.get("bgu"), act["dy"].float(),
                                n_heads=nh, gating_add=(gate_mode == 2), delta_scale=delta_scale,
                                x2_scale=x2_scale, gate_scale=gate_scale if has_gate else 1.0, has_gate=has_gate)
    x1 = act["x1"].to(dev).requires_grad_(True)
    x2 = act["x2"].to(dev).requires_grad_(True)
    rh = r // nh
    P = {k: t[k].to(dev).requires_grad_(True) for k in t if k not in ("x1", "x2", "dy")}
    dws = [P["wd"][i * rh:(i + 1) * rh].detach().clone().requires_grad_(True) for i in range(nh)]
    dbs = [P["bd"][i * rh:(i + 1) * rh].detach().clone().requires_grad_(True) for i in range(nh)]
    io = F._io_dtype(x2)
    tiles = max(F.rank_tiles(r), F.rank_tiles(rg) if has_gate else 1)
    pk_a = F.pack_pair(dws, dbs, P["wu"], P["bu"], io, tiles)
    pk_g = F.pack_pair([P["wgd"]], [P["bgd"]], P["wgu"], P["bgu"], io, tiles) if has_gate else None
    gp = (P["wgd"], P["bgd"], P["wgu"], P["bgu"]) if has_gate else None
    y = F.adapter_gate(x1, x2, dws, dbs, P["wu"], P["bu"], gp, pk_a, pk_g, gate_mode, delta_scale, x2_scale, gate_scale)
    y.backward(act["dy"].to(dev))
    torch.cuda.synchronize()
    errs = dict(y=rel_err(y, y_ref), dx2=rel_err(x2.grad, g_ref["x2"]))
    if has_gate:
        errs["dx1"] = rel_err(x1.grad, g_ref["x1"])
    errs["dwd"] = rel_err(torch.cat([w.grad for w in dws]), g_ref["wd"])
    errs["dbd"] = rel_err(torch.cat([b.grad for b in dbs]), g_ref["bd"])
    errs["dwu"] = rel_err(P["wu"].grad, g_ref["wu"])
    errs["dbu"] = rel_err(P["bu"].grad, g_ref["bu"])
    if has_gate:
        for k in ("wgd", "bgd", "wgu", "bgu"):
            errs["d" + k] = rel_err(P[k].grad, g_ref[k])
    if el_errs is not None:         # element-wise view of the input and weight gradients (el_rel_err)
        el_errs["y"] = el_rel_err(y, y_ref, K1_EL_FLOOR)
        el_errs["dx2"] = el_rel_err(x2.grad, g_ref["x2"], K1_EL_FLOOR)
        el_errs["dwd"] = el_rel_err(torch.cat([w.grad for w in dws]), g_ref["wd"], K1_EL_FLOOR)
        el_errs["dwu"] = el_rel_err(P["wu"].grad, g_ref["wu"], K1_EL_FLOOR)
        if has_gate:
            el_errs["dx1"] = el_rel_err(x1.grad, g_ref["x1"], K1_EL_FLOOR)
            el_errs["dwgd"] = el_rel_err(P["wgd"].grad, g_ref["wgd"], K1_EL_FLOOR)
            el_errs["dwgu"] = el_rel_err(P["wgu"].grad, g_ref["wgu"], K1_EL_FLOOR)
    if col_errs is not None:        # per-column view of the bias gradients (sums over all M rows)
        col_errs["dbd"] = col_rel_err(torch.cat([b.grad for b in dbs]), g_ref["bd"])
        col_errs["dbu"] = col_rel_err(P["bu"].grad, g_ref["bu"])
        if has_gate:
            col_errs["dbgd"] = col_rel_err(P["bgd"].grad, g_ref["bgd"])
            col_errs["dbgu"] = col_rel_err(P["bgu"].grad, g_ref["bgu"])
    return errs


def run_k2(dtype, M=224, d=768, r=96, scale=1.0, seed=1):
    import vlpet_amd.functional as F
    g = torch.Generator().manual_seed(seed)
    x, y, dy = _rand(g, M, d).to(dtype), _rand(g, M, d).to(dtype), _rand(g, M, d).to(dtype)
    W = dict(wd=_rand(g, r, d, scale=1 / math.sqrt(d)), bd=_rand(g, r, scale=0.1),
             wu=_rand(g, d, r, scale=1 / math.sqrt(r)), bu=_rand(g, d, scale=0.1))
    ref = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    xr, yr = x.float().clone().requires_grad_(True), y.float().clone().requires_grad_(True)
    out_ref = O.parallel_adapter(xr, yr, ref["wd"], ref["bd"], ref["wu"], ref["bu"], scale)
    out_ref.backward(dy.float())
    dev = "cuda"
    P = {k: v.to(dev).requires_grad_(True) for k, v in W.items()}
    xg, yg = x.detach().to(dev).requires_grad_(True), y.detach().to(dev).requires_grad_(True)
    pk = F.pack_pair([P["wd"]], [P["bd"]], P["wu"], P["bu"], F._io_dtype(xg))
    out = F.parallel_adapter(xg, yg, P["wd"], P["bd"], P["wu"], P["bu"], pk, scale)
    out.backward(dy.to(dev))
    torch.cuda.synchronize()
    errs = dict(out=rel_err(out, out_ref), dx=rel_err(xg.grad, xr.grad), dy=rel_err(yg.grad, yr.grad))
    for k in W:
        errs["d" + k] = rel_err(P[k].grad, ref[k].grad)
    return errs


def run_k3(dtype, M=200, d=768, r=8, alpha=32, p=0.0, seed=2, explicit_mask=False, rng_seed=0x5eed1234abcd):
    """K3 forward + backward vs the oracle.  Dropout parity is stated per mask (the reference's RNG stream cannot be
    matched): by default the mask comes from the kernels' generator, is exported by the forward and handed to the
    oracle; ``explicit_mask`` feeds a torch-drawn mask to both instead.  Returns the errors (+ ``keep_frac``)."""
    import vlpet_amd.functional as F
    g = torch.Generator().manual_seed(seed)
    x, dy = _rand(g, M, d).to(dtype), _rand(g, M, d).to(dtype)
    w, b = _rand(g, d, d, scale=1 / math.sqrt(d)), _rand(g, d, scale=0.1)
    A, B = _rand(g, r, d, scale=1 / math.sqrt(d)), _rand(g, d, r, scale=0.3)
    keep = (torch.rand(M, d, generator=g) >= p) if (p > 0 and explicit_mask) else None
    scaling = alpha / r
    dev = "cuda"
    xg = x.detach().to(dev).requires_grad_(True)
    Ag, Bg = A.to(dev).requires_grad_(True), B.to(dev).requires_grad_(True)
    base = torch.nn.functional.linear(xg.detach().float(), w.to(dev), b.to(dev)).to(dtype)
    pk = F.pack_pair([Ag], None, Bg, None, F._io_dtype(xg))
    out, mask = F.lora_delta(xg, base, Ag, Bg, pk, scaling, keep.to(dev).to(torch.uint8) if keep is not None else None,
                             p, rng_seed, return_mask=True)
    out.backward(dy.to(dev))
    torch.cuda.synchronize()
    if p > 0 and keep is None:
        keep = mask.cpu().bool()
    xr = x.float().clone().requires_grad_(True)
    Ar, Br = A.clone().requires_grad_(True), B.clone().requires_grad_(True)
    base_ref = torch.nn.functional.linear(xr, w, b)
    lora_ref = O.lora_linear(xr, torch.zeros_like(w), None, Ar, Br, scaling, keep, p)
    (base_ref.detach() + lora_ref).backward(dy.float())   # LoRA share of dx only
    out_ref = base_ref.detach() + lora_ref.detach()
    errs = dict(out=rel_err(out, out_ref), dx=rel_err(xg.grad, xr.grad), da=rel_err(Ag.grad, Ar.grad),
                db=rel_err(Bg.grad, Br.grad))
    if p > 0:
        errs["keep_frac"] = float(keep.float().mean())
    return errs


def run_pack_check(r=96, d=768, nh=4, fp32=False, seed=3):
    """bytes of the HIP pack kernel vs the numpy specification (tests/packing_spec.py)"""
    import vlpet_amd.functional as F
    import packing_spec as PK
    g = torch.Generator().manual_seed(seed)
    wd, bd = _rand(g, r, d), _rand(g, r)
    wu, bu = _rand(g, d, r), _rand(g, d)
    rh = r // nh
    dev = "cuda"
    io = 0 if fp32 else 1
    tiles = F.rank_tiles(r)
    pk = F.pack_pair([wd[i * rh:(i + 1) * rh].to(dev) for i in range(nh)],
                     [bd[i * rh:(i + 1) * rh].to(dev) for i in range(nh)], wu.to(dev), bu.to(dev), io, tiles)
    torch.cuda.synchronize()
    raw = pk.buf.cpu().numpy()
    NS = 2 if fp32 else 1
    nf = d // 16 * tiles
    pack_bytes = nf * NS * 1024
    wdp = np.zeros((32 * tiles, d), np.float32); wdp[:r] = wd.numpy()
    wup = np.zeros((d, 32 * tiles), np.float32); wup[:, :r] = wu.numpy()
    specs = [PK.pack_down4(wdp, NS), PK.pack_up4(wup, NS), PK.pack_up_t4(wup, NS), PK.pack_down_t4(wdp, NS)]
    bad = 0
    for i, spec in enumerate(specs):
        got = raw[i * pack_bytes:(i + 1) * pack_bytes].view(np.uint16).reshape(nf, NS, 512)
        hi, lo = PK.split_hi_lo(spec)
        bad += int((got[:, 0] != hi).sum())
        if fp32:
            bad += int((got[:, 1] != lo).sum())
    bias = raw[4 * pack_bytes:4 * pack_bytes + (32 * tiles + d) * 4].view(np.float32)
    exp = np.zeros(32 * tiles + d, np.float32); exp[:r] = bd.numpy(); exp[32 * tiles:] = bu.numpy()
    bad += int((bias != exp).sum())
    return bad
