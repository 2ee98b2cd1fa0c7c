"""Data-parallel Trainer on the GPU product path: two ranks (gloo, both on cuda:0 -- the test box has one GPU) with half
the batch each must land on the same parameters as one rank with the whole batch.  Exercises, under real backward
timing, the direct-write gradient sinks, the side-stream weight gradients, the bucketed all-reduce and the fused
optimizer's 1/world scaling."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _cfg_model():
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    cfg = HB.vlpet_config(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                          decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, vocab_size=500,
                          max_position_embeddings=64, feat_dim=128, adapter_down_dim=8, adapter_gating_down_dim=16,
                          decoder_enc_attn_value_parallel_adapter_down_dim=8, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0)
    torch.manual_seed(0)
    model = HB.VLBart(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    TR.trainable_names(model, cfg)
    model.train()
    return cfg, model


def _batches(cfg, n, order=("nlvr", "caption", "nlvr")):
    import vlpet_amd.train as TR
    gen = torch.Generator().manual_seed(5)
    return [TR.synthetic_batch(t, n, cfg, "cpu", gen) for t in order]   # token-mean losses: shard-size independent


GRAPH_ORDER = ("nlvr", "caption", "nlvr", "caption", "nlvr")      # per shape: eager step, capture + replay, replay


def _shard(b, rank, world):
    out = {}
    for k, v in b.items():
        if torch.is_tensor(v):
            out[k] = v.chunk(world)[rank].cuda()
        elif k == "vis_inputs":
            out[k] = tuple(t.chunk(world)[rank].cuda() for t in v)
        else:
            out[k] = v
    return out


def _worker(rank, world, port, out, graph=False):
    import vlpet_amd.train as TR
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg, model = _cfg_model()
    model.cuda()
    tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1, world_size=world, n_buckets=3,
                    overlap_wgrad=not graph,    # also covers the optional side-stream weight gradients
                    graph=graph)                # graph: captured forward + backward, the buckets reduced after the replay
    assert tr.graph == graph
    for b in _batches(cfg, 8, GRAPH_ORDER if graph else ("nlvr", "caption", "nlvr")):
        tr.step(_shard(b, rank, world))
    if graph:
        assert len(tr._graphs) == 2
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_two_gpu_ranks_equal_one_rank(tmp_path, graph):
    import vlpet_amd.train as TR
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, out, graph), nprocs=2, join=True)
    got = torch.load(out)
    cfg, model = _cfg_model()
    model.cuda()
    tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1)
    for b in _batches(cfg, 8, GRAPH_ORDER if graph else ("nlvr", "caption", "nlvr")):
        tr.step(_shard(b, 0, 1))
    worst = 0.0
    for n, p in model.named_parameters():
        if p.requires_grad:
            ref = p.detach().cpu()
            worst = max(worst, float((got[n] - ref).abs().max() / ref.abs().max().clamp_min(1e-3)))
    assert worst <= 2e-2, worst        # same bound as the GPU-vs-CPU trainer test (Adam amplifies rounding in the first steps)


def _rccl_worker(rank, world, port, out, graph=False, captured=False):
    import vlpet_amd.train as TR
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg, model = _cfg_model()
    model.cuda()
    tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1, world_size=1, n_buckets=3, force_collectives=True,
                    graph=graph, capture_collectives=captured)
    assert tr.capture_collectives == captured
    for b in _batches(cfg, 8, GRAPH_ORDER if graph else ("nlvr", "caption", "nlvr")):
        tr.step(_shard(b, 0, 1))
    torch.cuda.synchronize()
    if captured:        # the bucket all-reduces were launched from inside the captured backward and the captures succeeded (no eager fall-back)
        assert tr.graph and len(tr._graphs) == 2, (tr.graph, len(tr._graphs))
    torch.save({n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("graph,captured", [(False, False), (True, False), (True, True)], ids=["eager", "graph", "graph-captured-collectives"])
def test_rccl_collectives_on_one_rank(tmp_path, graph, captured):
    """backend "nccl" (= RCCL): the bucketed asynchronous all-reduces, the side-stream join and the fused optimizer on
    the real collective library, with a 1-rank communicator (the test box has one GPU).  Result == the plain run.
    graph-captured-collectives: Trainer(capture_collectives=True) -- the all-reduces leave from inside the captured backward and are
    replayed as graph nodes on the library's stream (the overlap of the eager path, kept under replay)."""
    import vlpet_amd.train as TR
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(1, port, out, graph, captured), nprocs=1, join=True)
    got = torch.load(out)
    cfg, model = _cfg_model()
    model.cuda()
    tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1)
    for b in _batches(cfg, 8, GRAPH_ORDER if graph else ("nlvr", "caption", "nlvr")):
        tr.step(_shard(b, 0, 1))
    for n, p in model.named_parameters():
        if p.requires_grad:
            ref = p.detach().cpu()
            assert float((got[n] - ref).abs().max() / ref.abs().max().clamp_min(1e-3)) <= 2e-2, n


def _rccl2_worker(rank, world, port, out):
    import vlpet_amd.train as TR
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    cfg, model = _cfg_model()
    model.cuda()
    tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1, world_size=world, n_buckets=3)
    for b in _batches(cfg, 8):
        tr.step({k: (v.cuda() if torch.is_tensor(v) else tuple(t.cuda() for t in v) if k == "vis_inputs" else v)
                 for k, v in _shard_cpu(b, rank, world).items()})
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad}, out)
    dist.destroy_process_group()


def _shard_cpu(b, rank, world):
    out = {}
    for k, v in b.items():
        if torch.is_tensor(v):
            out[k] = v.chunk(world)[rank]
        elif k == "vis_inputs":
            out[k] = tuple(t.chunk(world)[rank] for t in v)
        else:
            out[k] = v
    return out


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: real RCCL buckets over xGMI (the 1-GPU test box skips)")
def test_rccl_two_ranks_equal_one_rank(tmp_path):
    """Two ranks on two GPUs over RCCL (backend "nccl"), half the batch each: the bucketed asynchronous all-reduce of the flat
    trainable gradients + the fused optimizer's 1/world scaling land on the parameters of one rank with the whole batch.
    The first multi-GPU box that runs `pytest -m gpu` exercises the real collectives; one-GPU boxes skip."""
    import vlpet_amd.train as TR
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rccl2.pt")
    mp.spawn(_rccl2_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    cfg, model = _cfg_model()
    model.cuda()
    tr = TR.Trainer(model, cfg, lr=1e-2, total_steps=10, warmup_ratio=0.1)
    for b in _batches(cfg, 8):
        tr.step(_shard(b, 0, 1))
    for n, p in model.named_parameters():
        if p.requires_grad:
            ref = p.detach().cpu()
            assert float((got[n] - ref).abs().max() / ref.abs().max().clamp_min(1e-3)) <= 2e-2, n
