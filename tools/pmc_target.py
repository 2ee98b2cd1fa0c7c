#!/usr/bin/env python3
"""One op of the path launched a few times through the C ABI -- target of rocprofv3 --pmc / --kernel-trace passes (round 4).
usage: pmc_target.py <op> [M] [r]      op: k1fwd (training form), k1bwd, k2fwd, k2bwd, k3fwd (p = 0.1), k3bwd, k5fwd, k5bwd (p = 0.1), k4fwd (tiled GEMM form)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
op = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 28000
r = int(sys.argv[3]) if len(sys.argv) > 3 else (64 if op.startswith("k3") else 96)
d, dev, dt = 768, "cuda", torch.bfloat16
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x1, x2, dy = rn(M, d).to(dt), rn(M, d).to(dt), rn(M, d).to(dt)
out, dx1, dx2 = torch.empty_like(x1), torch.empty_like(x1), torch.empty_like(x1)
io, st = 1, torch.cuda.current_stream().cuda_stream
tiles = F.rank_tiles(r)
mk = lambda *s: rn(*s) * 0.05
f32 = dict(dtype=torch.float32, device=dev)
if op.startswith("k1"):
    W = [mk(r, d), mk(r), mk(d, r), mk(d), mk(r, d), mk(r), mk(d, r), mk(d)]
    pa, pg = F.pack_pair([W[0]], [W[1]], W[2], W[3], io, tiles), F.pack_pair([W[4]], [W[5]], W[6], W[7], io, tiles)
    sv = torch.empty(lib.vlpet_saved_bytes(M, tiles, io), dtype=torch.uint8, device=dev)
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 1, io); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    G = [torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32)] * 2
    G = [t.clone() for t in G]
    fwd = lambda: lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(), sv.data_ptr(),
                                                  M, d, tiles, 1, 1.0, 1.0, 1.0, io, st)
    # (from the forward's output y: the form the package runs since round 5; PMC_FROM_X2=1: the rounds 2-4 call)
    yp = None if os.environ.get("PMC_FROM_X2") else out.data_ptr()
    bwd = lambda: lib.vlpet_adapter_gate_bwd_saved_y(3, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), yp, sv.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(),
                                                     None, dx1.data_ptr(), dx2.data_ptr(), *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws, M, d, tiles, 1,
                                                     1.0, 1.0, 1.0, io, st)
elif op.startswith("k2"):
    wd, bd, wu, bu = mk(r, d), mk(r), mk(d, r), mk(d)
    pk = F.pack_pair([wd], [bd], wu, bu, io, tiles)
    sv = torch.empty(lib.vlpet_saved_bytes(M, tiles, io), dtype=torch.uint8, device=dev)
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 0, io); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    G = [torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32)]
    fwd = lambda: lib.vlpet_parallel_adapter_fwd_save(x1.data_ptr(), x2.data_ptr(), pk.buf.data_ptr(), out.data_ptr(), sv.data_ptr(), M, d, tiles, 1.0, io, st)
    bwd = lambda: lib.vlpet_parallel_adapter_bwd_saved(dy.data_ptr(), x1.data_ptr(), sv.data_ptr(), pk.buf.data_ptr(), dx1.data_ptr(),
                                                       *[t.data_ptr() for t in G], r, ws.data_ptr(), nws, M, d, tiles, 1.0, io, st)
elif op.startswith("k3"):
    A, B = mk(r, d), mk(d, r)
    pk = F.pack_pair([A], None, B, None, io)
    tiles = pk.tiles
    sv = torch.empty(lib.vlpet_lora_saved_bytes(M, d, tiles, io), dtype=torch.uint8, device=dev)
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 0, io); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    da, db = torch.empty(r, d, **f32), torch.empty(d, r, **f32)
    fwd = lambda: lib.vlpet_lora_delta_fwd_save(x1.data_ptr(), x2.data_ptr(), pk.buf.data_ptr(), None, 0.1, 1234, None, out.data_ptr(), sv.data_ptr(),
                                                M, d, tiles, 0.5, io, st)
    bwd = lambda: lib.vlpet_lora_delta_bwd_saved(dy.data_ptr(), x1.data_ptr(), sv.data_ptr(), pk.buf.data_ptr(), None, 0.1, 1234, dx1.data_ptr(),
                                                 da.data_ptr(), db.data_ptr(), r, ws.data_ptr(), nws, M, d, tiles, 0.5, io, st)
elif op.startswith("k4"):          # K4 forward, tiled-GEMM form (csrc/visproj_gemm.hip), 2048 -> 768
    Fd = 2048
    feats = rn(M, Fd).to(dt); wv = (rn(d, Fd) * 0.02).to(dt); bv = rn(d) * 0.1
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    Rr = rn(M, d).to(dt); xh = torch.empty_like(out); rstd = torch.empty(M, device=dev)
    nwsg = lib.vlpet_visproj_gemm_workspace_bytes(M, Fd, d); wsg = torch.zeros(nwsg, dtype=torch.uint8, device=dev)
    fwd = lambda: lib.vlpet_visproj_fwd_gemm(feats.data_ptr(), wv.data_ptr(), bv.data_ptr(), gam.data_ptr(), bet.data_ptr(), Rr.data_ptr(), out.data_ptr(),
                                             xh.data_ptr(), rstd.data_ptr(), None, wsg.data_ptr(), nwsg, M, Fd, d, 1e-5, 0, io, st)
    bwd = fwd
else:
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    part = torch.empty(lib.vlpet_sublayer_tail_partials(M), 2, d, device=dev)
    fwd = lambda: lib.vlpet_sublayer_tail_fwd(x2.data_ptr(), x1.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(), None, mean.data_ptr(),
                                              rstd.data_ptr(), None, M, d, 1e-5, 0.1, 7, 1, io, st)
    bwd = lambda: lib.vlpet_sublayer_tail_bwd_out(dy.data_ptr(), out.data_ptr(), rstd.data_ptr(), gam.data_ptr(), bet.data_ptr(), dx1.data_ptr(),
                                                  dx2.data_ptr(), part.data_ptr(), M, d, 0.1, 7, io, st)
assert fwd() == 0
fn = bwd if op.endswith("bwd") else fwd
for _ in range(5):
    assert fn() == 0
torch.cuda.synchronize()
