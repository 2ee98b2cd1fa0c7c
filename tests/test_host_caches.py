"""Derived copies of frozen tensors: edits through ``.data`` do not bump ``Tensor._version`` of the parameter,
so the caches carry ``functional.FROZEN_EPOCH`` and the hosts invalidate them after ``load_state_dict``."""
import torch

import vlpet_amd.functional as VF
from vlpet_amd import tail


def test_frozen_copy_follows_invalidate_caches():
    w = torch.nn.Parameter(torch.ones(8, dtype=torch.bfloat16), requires_grad=False)
    a = tail._f32_frozen(w)
    assert float(a.sum()) == 8.0
    w.data.mul_(2)                                  # invisible to w._version
    stale = tail._f32_frozen(w)
    VF.invalidate_caches()
    fresh = tail._f32_frozen(w)
    assert float(fresh.sum()) == 16.0 and (stale is a or float(stale.sum()) == 16.0)


def test_load_state_dict_invalidates():
    import vlpet_amd.host.bart as HB
    e0 = VF.FROZEN_EPOCH
    cfg = HB.vlpet_config(d_model=64, encoder_layers=1, decoder_layers=1, encoder_attention_heads=4, decoder_attention_heads=4,
                          encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=128, feat_dim=64, adapter_down_dim=8,
                          adapter_gating_down_dim=8)
    m = HB.VLBart(cfg)
    m.load_state_dict(m.state_dict())
    assert VF.FROZEN_EPOCH > e0
