"""K4's position / order branch (csrc/vispos.hip) through the C ABI at the configs[1] row counts: time per launch, against the
library-op chain the host used before round 6 (visual._position_terms + autograd)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vlpet_amd import _lib  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _lib.load()
    dev = "cuda"
    d, N, V, n_img = 768, 36, 50465, 2
    st = torch.cuda.current_stream().cuda_stream
    for B, Nn in ((500, 36), (833, 36), (166, 72), (416, 36), (63, 36)):
        M = B * Nn
        pos = torch.rand(B, Nn, 4, device=dev)
        w, b, g, be = torch.randn(d, 5, device=dev), torch.randn(d, device=dev), torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
        img_t, obj_t = torch.randn(n_img, d, device=dev), torch.randn(V, d, device=dev).bfloat16()
        ids = torch.randint(0, n_img, (B, Nn), device=dev) if Nn == 72 else None
        out = torch.empty(B, Nn, d, dtype=torch.bfloat16, device=dev)
        dout = torch.randn(B, Nn, d, device=dev).bfloat16()
        dw, db, dg, dbe, dimg = (torch.empty(d, 5, device=dev), torch.empty(d, device=dev), torch.empty(d, device=dev),
                                 torch.empty(d, device=dev), torch.empty(n_img, d, device=dev))
        nws = lib.vlpet_vispos_bwd_workspace_bytes(M, d, n_img)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        P = lambda t: None if t is None else t.data_ptr()
        fwd = lambda: lib.vlpet_vispos_fwd(P(pos), P(w), P(b), P(g), P(be), P(img_t), _lib.VLPET_F32, n_img, P(ids), Nn if ids is not None else 0,
                                           P(obj_t), _lib.VLPET_BF16, V, None, 0, P(out), M, Nn, d, 1e-5, 0, _lib.VLPET_BF16, st)
        bwd = lambda: lib.vlpet_vispos_bwd(P(dout), P(pos), P(w), P(b), P(g), n_img, P(ids), Nn if ids is not None else 0, P(dw), P(db), P(dg), P(dbe),
                                           P(dimg), P(ws), nws, M, Nn, d, 1e-5, 0, _lib.VLPET_BF16, st)
        assert fwd() == 0 and bwd() == 0
        tf, tb = timed(fwd), timed(bwd)
        # the library-op chain
        lin = torch.nn.Linear(5, d).to(dev)
        ln = torch.nn.LayerNorm(d).to(dev)
        img_e = torch.nn.Embedding(n_img, d).to(dev)
        obj_ids = (V - 1 - torch.arange(Nn, device=dev)).unsqueeze(0)
        img_ids = ids if ids is not None else torch.zeros(1, Nn, dtype=torch.long, device=dev)

        def chain():
            p5 = torch.cat([pos, ((pos[:, :, 3] - pos[:, :, 2]) * (pos[:, :, 1] - pos[:, :, 0])).unsqueeze(2)], dim=2)
            R = ln(lin(p5)) + img_e(img_ids).float() + obj_t[obj_ids].float()
            Rb = R.expand(B, Nn, d).to(torch.bfloat16)
            Rb.backward(dout)
        tc = timed(chain, 20)
        print(f"vispos M={M:6d} (B={B}, N={Nn}): fwd {tf:6.1f} us ({M * d * 2 / tf / 1e6:5.2f} TB/s written)   bwd + finalize {tb:6.1f} us "
              f"({M * d * 2 / tb / 1e6:5.2f} TB/s read)   library-op chain fwd + bwd {tc:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
