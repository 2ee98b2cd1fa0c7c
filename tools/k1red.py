#!/usr/bin/env python3
"""K1 backward: in-launch reduce-scatter of the row-chunk partials (round 6, csrc/cols_reduce.h) against the round-3 form (partial slabs +
wgrad_finalize_kernel, ABI phases bit 5), same process, same buffers, warm and cold (a 1 GiB read-modify-write before every timed call).
usage: k1red.py M [M ...]        K1RED_R=8|32|96 (bottleneck)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
from kbench import timeit


def run(M):
    dt, r, d, dev = torch.bfloat16, int(os.environ.get("K1RED_R", "96")), 768, "cuda"
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    x1, x2, dy, dxin = (torch.randn(M, d, device=dev, generator=g).to(dt) for _ in range(4))
    mk = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
    W = [mk(r, d), mk(r), mk(d, r), mk(d), mk(r, d), mk(r), mk(d, r), mk(d)]
    io, tiles = F._io_dtype(x2), F.rank_tiles(r)
    pa = F.pack_pair([W[0]], [W[1]], W[2], W[3], io, tiles); pg = F.pack_pair([W[4]], [W[5]], W[6], W[7], io, tiles)
    out = torch.empty_like(x2)
    st = torch.cuda.current_stream().cuda_stream
    nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 1, io)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    G = [torch.empty_like(w) for w in W]
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    sv = torch.empty(lib.vlpet_saved_bytes(M, tiles, io), dtype=torch.uint8, device=dev)
    assert lib.vlpet_adapter_gate_fwd_save(x1.data_ptr(), x2.data_ptr(), pa.buf.data_ptr(), pg.buf.data_ptr(), out.data_ptr(), sv.data_ptr(),
                                           M, d, tiles, 1, 1.0, 1.0, 1.0, io, st) == 0

    def bwd(ph):
        def f():
            rc = lib.vlpet_adapter_gate_bwd_saved_y(ph, dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), out.data_ptr(), sv.data_ptr(), pa.buf.data_ptr(),
                                                    pg.buf.data_ptr(), dxin.data_ptr(), dx1.data_ptr(), dx2.data_ptr(), *[t.data_ptr() for t in G], r, r,
                                                    ws.data_ptr(), nws, M, d, tiles, 1, 1.0, 1.0, 1.0, io, st)
            assert rc == 0
        return f
    evict = torch.zeros(1 << 28, dtype=torch.float32, device=dev)

    def cold(fn, iters=15):
        ts = []
        for _ in range(iters):
            evict.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    w = lambda fn: min(timeit(fn, iters=60, warm=5) for _ in range(3))
    # ABBA
    a1, b1, b2, a2 = w(bwd(3)), w(bwd(3 | 32)), w(bwd(3 | 32)), w(bwd(3))
    p1 = w(bwd(1))
    ca1, cb1, cb2, ca2 = cold(bwd(3)), cold(bwd(3 | 32)), cold(bwd(3 | 32)), cold(bwd(3))
    frac = lambda t: 5 * d * M * 2 / t / 1e3 / 8000
    new, old, cnew, cold_ = min(a1, a2), min(b1, b2), min(ca1, ca2), min(cb1, cb2)
    print(f"k1red M={M:6d} r={r:3d}: pass 1 {p1:5.1f} us | whole op warm: in-launch reduce {a1:6.1f} / {a2:6.1f}, finalize launch {b1:6.1f} / {b2:6.1f} us "
          f"-> {new:6.1f} vs {old:6.1f} ({frac(new):.3f} vs {frac(old):.3f}) | cold: {ca1:6.1f} / {ca2:6.1f} vs {cb1:6.1f} / {cb2:6.1f} -> {cnew:6.1f} vs {cold_:6.1f} "
          f"({frac(cnew):.3f} vs {frac(cold_):.3f})", flush=True)


if __name__ == "__main__":
    for M in [int(a) for a in sys.argv[1:]] or [28000]:
        run(M)
