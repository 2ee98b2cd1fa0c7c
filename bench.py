#!/usr/bin/env python3
"""bench.py -- multitask fine-tuning throughput of the MI355X-native VL-PET path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): BART-base + VL-PET-large (r = r_g = dec r = 96, N_h = 4),
image-text multitask, bf16 activations / frozen weights, fp32 trainable masters.  One step = one full
train step (forward, backward, gradient exchange, clip 5.0, AdamW) on one task batch; steps cycle
vqa -> gqa -> nlvr -> caption with the reference's per-task batch sizes (500 / 833 / 166 / 416,
multitask.py:682-695) per GPU (weak scaling).  Synthetic CLIP-feature + token batches are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s measured achievable)
# HBM-side bytes per row of pet_bwd_kernel<bf16,3,gate> from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE;
# profiles/r01_pmc_traffic_k1_bwd.md).  Collected with rocprofv3 --pmc in its own run, not inside this script.
PMC_TRAFFIC_BYTES_PER_ROW = {"k1_bwd_rows": 12175.0}
TASK_ORDER = ["vqa", "gqa", "nlvr", "caption"]


def cpu_baseline(steps=60, warm=3, batch=4):
    """Reference path restated on the host CPU (kind "port"): the same host model with the PET ops
    routed to oracle/vlpet_oracle.py (plain eager PyTorch in the reference's op order), full train step,
    fp32, BASELINE.json configs[0] (VQA, batch 4)."""
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from oracle.host_patch import cpu_reference_ops

    torch.set_num_threads(min(16, os.cpu_count() or 1))      # small-batch eager ops do not scale past ~16 threads
    with cpu_reference_ops():
        torch.manual_seed(1234)
        cfg = HB.vlpet_config()
        model = HB.VLBart(cfg)
        TR.trainable_names(model, cfg)
        model.train()
        tr = TR.Trainer(model, cfg, total_steps=1000)
        gen = torch.Generator().manual_seed(1234)
        b = TR.synthetic_batch("vqa", batch, cfg, "cpu", gen)
        for _ in range(warm):
            tr.step(b)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(b)
        dt = time.perf_counter() - t0
    return dict(value=round(batch * steps / dt, 3), unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"configs[0]: BART-base VL-PET-large r=96, VQA batch {batch}, S=20+36, fp32, full train step "
                       f"(fwd+bwd+clip+AdamW) through oracle/vlpet_oracle.py on the host CPU, {warm} warm-up + "
                       f"{steps} timed steps ({dt:.1f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=500, help="per-GPU VQA batch (other tasks scale like the reference)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--buckets", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to exercise the DP path on a 1-GPU box")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="K1 weight gradients on a side stream (measured 5 %% SLOWER on one MI355X: the step is GPU-bound)")
    ap.add_argument("--model", default="bart", choices=["bart", "t5"],
                    help="bart = BASELINE configs[1] (the headline line); t5 = configs[2] (T5-base, r = r_g = 192, --batch 300)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    import vlpet_amd.functional as VF
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from vlpet_amd import _lib
    _lib.load()     # fail loudly before anything is timed

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(1234)      # same initial state on every rank: no parameter broadcast needed
    if args.model == "t5":
        import vlpet_amd.host.t5 as HT
        cfg = HT.vlt5_config()
        model = HT.VLT5(cfg)
    else:
        cfg = HB.vlpet_config()
        model = HB.VLBart(cfg)
    names = TR.trainable_names(model, cfg)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    model.to(dev)
    TR.cast_frozen(model, dtype)
    model.train()
    total_steps = max(args.steps + args.warmup, 10)
    tr = TR.Trainer(model, cfg, lr=1e-3, clip=5.0, total_steps=total_steps, world_size=world, n_buckets=args.buckets,
                    overlap_wgrad=args.overlap_wgrad)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    batches = {t: TR.synthetic_batch(t, TR.TASK_BATCH[t](args.batch), cfg, dev, gen) for t in TASK_ORDER}
    order = [TASK_ORDER[i % 4] for i in range(args.warmup + args.steps)]

    for i in range(args.warmup):
        tr.step(batches[order[i]])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    VF.TIMER = VF.KernelTimer()
    t0 = time.perf_counter()
    samples = 0
    for i in range(args.warmup, args.warmup + args.steps):
        b = batches[order[i]]
        tr.step(b)
        samples += b["input_ids"].shape[0]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer, VF.TIMER = VF.TIMER, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([samples], device=dev, dtype=torch.float64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        samples = int(s.item())

    if rank == 0:
        esz = 2 if dtype == torch.bfloat16 else 4
        d = cfg.d_model
        agg = timer.summary()
        # algorithmic bytes per row: fwd reads x1, x2, writes y; bwd rows reads dy, x1, x2, writes dx1, dx2
        # K5 tail: fwd reads y, x1, writes out; bwd reads dout, writes dx1, dy (the saved pre-norm sum is extra traffic)
        per_row = {"k1_fwd": 3 * d * esz, "k1_bwd_rows": 5 * d * esz, "k1_bwd_wgrad": 0, "k2_fwd": 3 * d * esz,
                   "k2_bwd": 3 * d * esz, "k5_fwd": 3 * d * esz, "k5_bwd": 3 * d * esz}
        kernels = {}
        for name, a in agg.items():
            by = per_row.get(name, 0) * a["rows"]
            kernels[name] = dict(launches=a["launches"], avg_us=round(a["total_us"] / a["launches"], 2),
                                 total_ms=round(a["total_us"] / 1e3, 3),
                                 algorithmic_GBps=round(by / a["total_us"] / 1e3, 1) if by else None)
        # dominant HIP kernel of the hot path by time: the row-parallel K1 backward
        dom = "k1_bwd_rows" if "k1_bwd_rows" in agg else "k1_fwd"
        a = agg[dom]
        achieved = per_row[dom] * a["rows"] / a["total_us"] / 1e3     # GB/s
        tiles = 3 if args.model == "bart" else 6
        # (r = 96: the chain-split kernel pet_gate_bwd2.hip with the forward's saved activations; r = 192: pet_bwd.hip)
        roof = dict(bound="hbm", kernel={"k1_bwd_rows": (f"pet_gate_bwd2_kernel<{args.dtype},{tiles}>" if tiles <= 3
                                                         else f"pet_bwd_kernel<{args.dtype},{tiles},gate>"),
                                         "k1_fwd": f"pet_gate_fwd_kernel<{args.dtype},{tiles}>"}[dom],
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                    traffic=(round(PMC_TRAFFIC_BYTES_PER_ROW[dom] * a["rows"] / a["launches"])
                             if dom in PMC_TRAFFIC_BYTES_PER_ROW and args.dtype == "bf16" and args.model == "bart" else None),
                    traffic_source="profiles/r01_pmc_traffic_k1_bwd.md (PMC passes at M=28000, scaled by rows per launch)",
                    avg_launch_us=round(a["total_us"] / a["launches"], 2),
                    avg_rows_per_launch=round(a["rows"] / a["launches"], 1),
                    algorithmic_bytes_per_row=per_row[dom])
        out = {
            "metric": "multitask samples/sec (BART-base, r=96)" if args.model == "bart" else "multitask samples/sec (T5-base, r=192)",
            "value": round(samples / dt, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("configs[1]: BART-base + VL-PET-large (r=96)" if args.model == "bart" else
                                    "configs[2]: T5-base + VL-PET-large (r=192, gate scale 0.3)") +
                                   " image-text multitask, full train step (fwd+bwd+grad exchange+clip+AdamW), random-init weights",
                       "per_gpu_task_batch": {t: TR.TASK_BATCH[t](args.batch) for t in TASK_ORDER},
                       "enc_rows_per_step": {t: TR.TASK_BATCH[t](args.batch) * (TR.TEXT_LEN[t] + (72 if t == "nlvr" else 36))
                                             for t in TASK_ORDER},
                       "trainable_params": n_train, "parallelism": f"dp{world}"},
            "roofline": roof, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
