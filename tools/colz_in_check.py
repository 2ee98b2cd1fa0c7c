#!/usr/bin/env python3
"""dx1_in form of the gated K1 backward (vlpet_adapter_gate_bwd_saved_acc) against the plain form + an explicit add, per tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
lib = _lib.load()
for M in [int(a) for a in sys.argv[1:]] or [1000, 8232, 28000]:
    d, r, dev, dtype = 768, 96, "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(11)
    x1, x2, dy, dxin = (torch.randn(M, d, device=dev, generator=g).to(dtype) for _ in range(4))
    mk = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
    W = [mk(r, d), mk(r), mk(d, r), mk(d), mk(r, d), mk(r), mk(d, r), mk(d)]
    io, tiles = F._io_dtype(x2), F.rank_tiles(r)
    pa = F.pack_pair([W[0]], [W[1]], W[2], W[3], io, tiles); pg = F.pack_pair([W[4]], [W[5]], W[6], W[7], io, tiles)
    st = torch.cuda.current_stream().cuda_stream
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 1, io)
    out = torch.empty_like(x2)
    sv = torch.empty(lib.vlpet_saved_bytes(M, tiles, io), dtype=torch.uint8, device=dev)
    assert lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(), sv.data_ptr(), M, d, tiles, 1, 1.0, 1.0, 0.7, io, st) == 0
    res = []
    for acc in (False, True):
        dx1 = torch.zeros_like(x1); dx2 = torch.zeros_like(x2)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        G = [torch.empty_like(w) for w in W]
        common = [t.data_ptr() for t in G] + [r, r, ws.data_ptr(), nws, M, d, tiles, 1, 1.0, 1.0, 0.7, io, st]
        if acc:
            rc = lib.vlpet_adapter_gate_bwd_saved_acc(3, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), sv.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), dxin.data_ptr(), dx1.data_ptr(), dx2.data_ptr(), *common)
        else:
            rc = lib.vlpet_adapter_gate_bwd_saved(3, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), sv.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), dx1.data_ptr(), dx2.data_ptr(), *common)
        assert rc == 0
        torch.cuda.synchronize()
        res.append([dx1.float(), dx2.float()] + [t.float() for t in G])
    ref = res[0][0] + dxin.float()
    err = (res[1][0] - ref).abs()
    rows = (err.max(dim=1).values > 1e-2 * ref.abs().max()).nonzero().flatten()
    print(f"M={M}: dx1 err {err.max().item() / ref.abs().max().item():.2e}; bad rows {rows.numel()} first {rows[:8].tolist()} last {rows[-4:].tolist()}; others equal: {[torch.equal(a, b) for a, b in zip(res[0][1:], res[1][1:])]}", flush=True)
    if rows.numel():
        r0 = int(rows[0]); bad_cols = (err[r0] > 1e-2 * ref.abs().max()).nonzero().flatten()
        print("   row", r0, "bad cols", bad_cols.numel(), bad_cols[:8].tolist())
