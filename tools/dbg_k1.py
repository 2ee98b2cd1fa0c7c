import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, gpu_cases as C
from oracle import vlpet_oracle as O
import vlpet_amd.functional as F
M, d, r, rg, nh = 128, 128, 16, 16, 2
t = C.make_k1(0, M, d, r, rg, nh)
for dtype in (torch.float32,):
    act = {k: t[k].to(dtype) for k in ("x1", "x2", "dy")}
    y_ref, _ = O.k1_fwd_bwd(act["x1"].float(), act["x2"].float(), t["wd"], t["bd"], t["wu"], t["bu"], t["wgd"], t["bgd"], t["wgu"], t["bgu"], act["dy"].float(), n_heads=nh, gating_add=False, delta_scale=1.0, x2_scale=1.0, gate_scale=1.0, has_gate=True)
    dev = "cuda"
    x1 = act["x1"].to(dev); x2 = act["x2"].to(dev)
    P = {k: t[k].to(dev) for k in t if k not in ("x1", "x2", "dy")}
    rh = r // nh
    dws = [P["wd"][i*rh:(i+1)*rh].contiguous() for i in range(nh)]; dbs = [P["bd"][i*rh:(i+1)*rh].contiguous() for i in range(nh)]
    io = F._io_dtype(x2); tiles = F.rank_tiles(r)
    pk_a = F.pack_pair(dws, dbs, P["wu"], P["bu"], io, tiles); pk_g = F.pack_pair([P["wgd"]], [P["bgd"]], P["wgu"], P["bgu"], io, tiles)
    with torch.no_grad():
        y = F.adapter_gate(x1, x2, dws, dbs, P["wu"], P["bu"], (P["wgd"], P["bgd"], P["wgu"], P["bgu"]), pk_a, pk_g, 1, 1.0, 1.0, 1.0)
    e = (y.float().cpu() - y_ref).abs()
    print(dtype, "max", e.max().item())
    print("by 32-col block:", [round(e[:, c:c+32].max().item(), 4) for c in range(0, d, 32)])
    print("by 16-col block row0-31:", [round(e[:32, c:c+16].max().item(), 4) for c in range(0, d, 16)])
    print("by 32-row block:", [round(e[r0:r0+32].max().item(), 4) for r0 in range(0, M, 32)])
    # what does y look like vs pieces
    lin_only = None
    import torch.nn.functional as Fn
    def gelu_new(x): return 0.5 * x * (1 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))
    X1, X2 = act["x1"].float(), act["x2"].float()
    delta = gelu_new(X2 @ t["wd"].T + t["bd"]) @ t["wu"].T + t["bu"]
    gpre = gelu_new(X1 @ t["wgd"].T + t["bgd"]) @ t["wgu"].T + t["bgu"]
    g = torch.sigmoid(gpre)
    Y = y.float().cpu()
    FE = 32
    def roll(a, k): return torch.roll(a, k * FE, dims=1)
    cands = {"ref": (X2 + delta) * g}
    for k in (-2, -1, 1, 2):
        cands[f"g_roll{k}"] = (X2 + delta) * roll(g, k)
        cands[f"x2_roll{k}"] = (roll(X2, k) + delta) * g
        cands[f"delta_roll{k}"] = (X2 + roll(delta, k)) * g
        cands[f"x2g_roll{k}"] = (roll(X2, k) + delta) * roll(g, k)
    cands["nogate"] = X2 + delta
    cands["delta*g"] = delta * g
    cands["x2*g"] = X2 * g
    cands["x1res"] = (X1 + delta) * g
    for k, v in cands.items():
        print(k, round((Y - v).abs().max().item(), 4), "cols32-63:", round((Y - v)[:, 32:64].abs().max().item(), 4))
    R = Y / g - delta
    torch.set_printoptions(precision=3, linewidth=200)
    print("R[0,:40]", R[0, :40]); print("X2[0,:40]", X2[0, :40])
    # find for R[0, 0:4] the best match anywhere in X2 / X1
    for name, T in (("X2", X2), ("X1", X1), ("g", g), ("delta", delta)):
        dd = (T.unsqueeze(-1) - R[0, 0]).abs()
        idx = (T - R[0, 0]).abs().argmin(); print(name, "closest to R[0,0]:", divmod(idx.item(), T.shape[1]), T.flatten()[idx].item(), R[0, 0].item())
