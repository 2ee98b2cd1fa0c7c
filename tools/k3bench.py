#!/usr/bin/env python3
"""K3 (LoRA delta) forward / backward timings through the C ABI with HIP events: ranks 8 / 64 / 128, without dropout,
with the in-kernel generator (p = 0.1) and with an explicit byte mask.  usage: tools/k3bench.py [M] [dtype]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlpet_amd.functional as F
from vlpet_amd import _lib
from kbench import timeit


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 28000
    dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
    d, dev = 768, "cuda"
    esz = 2 if dt == torch.bfloat16 else 4
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, d, device=dev, generator=g).to(dt)
    base = torch.randn(M, d, device=dev, generator=g).to(dt)
    dy = torch.randn(M, d, device=dev, generator=g).to(dt)
    keep = (torch.rand(M, d, device=dev, generator=g) >= 0.1).to(torch.uint8)
    out, dx = torch.empty_like(x), torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    io = F._io_dtype(x)
    by = 3 * d * M * esz
    print(f"M={M} dtype={dt}: algorithmic bytes fwd = bwd = {by/1e6:.1f} MB (3*d*M*b)")
    for r in (8, 64, 128):
        A = torch.randn(r, d, device=dev, generator=g) * 0.05
        B = torch.randn(d, r, device=dev, generator=g) * 0.05
        pk = F.pack_pair([A], None, B, None, io)
        tiles = pk.tiles
        nws = lib.vlpet_bwd_workspace_bytes(M, d, tiles, 0, io)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        da, db = torch.empty(r, d, device=dev), torch.empty(d, r, device=dev)
        sv = torch.empty(lib.vlpet_lora_saved_bytes(M, d, tiles, io), dtype=torch.uint8, device=dev)
        for label, km, p in (("no dropout", None, 0.0), ("generator p=0.1", None, 0.1), ("byte mask p=0.1", keep, 0.1)):
            kp = km.data_ptr() if km is not None else None
            def fwd():
                rc = lib.vlpet_lora_delta_fwd(x.data_ptr(), base.data_ptr(), pk.buf.data_ptr(), kp, p, 1234, None, out.data_ptr(),
                                              M, d, tiles, 0.5, io, st); assert rc == 0
            def bwd():
                rc = lib.vlpet_lora_delta_bwd(dy.data_ptr(), x.data_ptr(), pk.buf.data_ptr(), kp, p, 1234, dx.data_ptr(), da.data_ptr(),
                                              db.data_ptr(), r, ws.data_ptr(), nws, M, d, tiles, 0.5, io, st); assert rc == 0
            def fwd_s():        # training form: z and the packed mask are left for the backward
                rc = lib.vlpet_lora_delta_fwd_save(x.data_ptr(), base.data_ptr(), pk.buf.data_ptr(), kp, p, 1234, None, out.data_ptr(),
                                                   sv.data_ptr(), M, d, tiles, 0.5, io, st); assert rc == 0
            def bwd_s():
                rc = lib.vlpet_lora_delta_bwd_saved(dy.data_ptr(), x.data_ptr(), sv.data_ptr(), pk.buf.data_ptr(), kp, p, 1234, dx.data_ptr(),
                                                    da.data_ptr(), db.data_ptr(), r, ws.data_ptr(), nws, M, d, tiles, 0.5, io, st); assert rc == 0
            def fwd_r8(save):   # rank <= 8: the streaming row kernel (csrc/lora8.hip), inference / training form
                def f():
                    rc = lib.vlpet_lora_delta_fwd_r8(x.data_ptr(), base.data_ptr(), pk.buf.data_ptr(), kp, p, 1234, None, out.data_ptr(),
                                                     sv.data_ptr() if save else None, M, d, r, 0.5, io, st); assert rc == 0
                return f
            tf, tb = timeit(fwd), timeit(bwd)
            tfs = timeit(fwd_s); tbs = timeit(bwd_s)
            if lib.vlpet_lora_r8_applies(M, d, r, io):
                t8, t8s = timeit(fwd_r8(False)), timeit(fwd_r8(True))
                print(f"r={r:4d}         {label:18s}: streaming form (lora8): fwd {t8:7.1f} us (frac {by/t8/1e3/8000:.3f})   training form: fwd {t8s:7.1f} us "
                      f"(frac {by/t8s/1e3/8000:.3f})")
            print(f"r={r:4d} tiles={tiles} {label:18s}: fwd {tf:7.1f} us ({by/tf/1e3:7.1f} GB/s, frac {by/tf/1e3/8000:.3f})   "
                  f"bwd rows+wgrad {tb:7.1f} us ({by/tb/1e3:7.1f} GB/s, frac {by/tb/1e3/8000:.3f})   | training form: fwd {tfs:7.1f} us "
                  f"(frac {by/tfs/1e3/8000:.3f}), bwd {tbs:7.1f} us (frac {by/tbs/1e3/8000:.3f})")


if __name__ == "__main__":
    main()
