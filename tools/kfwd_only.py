#!/usr/bin/env python3
"""Launch only the K1 forward ('fwd'; 'fwds' = training form that saves z / gelu') or the backward rows + weight-gradient
kernels ('bwd' = with the saved activations, what training runs; 'bwdr' = recompute form) a few times -- target for
rocprofv3 --pmc / --kernel-trace runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 28000
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
d, r, dev, dt = 768, 96, "cuda", torch.bfloat16
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
x1 = torch.randn(M, d, device=dev, generator=g).to(dt); x2 = torch.randn(M, d, device=dev, generator=g).to(dt)
dy = torch.randn(M, d, device=dev, generator=g).to(dt)
mk = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
W = [mk(r, d), mk(r), mk(d, r), mk(d), mk(r, d), mk(r), mk(d, r), mk(d)]
io, tiles = 1, 3
pa = F.pack_pair([W[0]], [W[1]], W[2], W[3], io, tiles); pg = F.pack_pair([W[4]], [W[5]], W[6], W[7], io, tiles)
out = torch.empty_like(x2); st = torch.cuda.current_stream().cuda_stream
nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 1, io); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
f32 = dict(dtype=torch.float32, device=dev)
G = [torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32),
     torch.empty(r, d, **f32), torch.empty(r, **f32), torch.empty(d, r, **f32), torch.empty(d, **f32)]
dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
nsv = lib.vlpet_saved_bytes(M, tiles, io); sv = torch.empty(nsv, dtype=torch.uint8, device=dev)
assert lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(), sv.data_ptr(), M, d, tiles, 1, 1.0, 1.0, 1.0, io, st) == 0
for _ in range(5):
    if mode == "fwds":
        assert lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(), sv.data_ptr(), M, d, tiles, 1, 1.0, 1.0, 1.0, io, st) == 0
    elif mode == "bwd":
        assert lib.vlpet_adapter_gate_bwd_saved(3, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), sv.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), dx1.data_ptr(), dx2.data_ptr(),
                                                *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws, M, d, tiles, 1, 1.0, 1.0, 1.0, io, st) == 0
    elif mode == "fwd":
        assert lib.vlpet_adapter_gate_fwd(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(), M, d, tiles, 1, 1.0, 1.0, 1.0, io, st) == 0
    else:
        assert lib.vlpet_adapter_gate_bwd(dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), dx1.data_ptr(), dx2.data_ptr(),
                                          *[t.data_ptr() for t in G], r, r, ws.data_ptr(), nws, M, d, tiles, 1, 1.0, 1.0, 1.0, io, st) == 0
torch.cuda.synchronize()
