#!/usr/bin/env python3
"""bench.py -- multitask fine-tuning throughput of the MI355X-native VL-PET path.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

Workload (default, BASELINE.json configs[1]): BART-base + VL-PET-large (r = r_g = dec r = 96, N_h = 4),
image-text multitask, bf16 activations / frozen weights, fp32 trainable masters.  One step = one full
train step (forward, backward, gradient exchange, clip 5.0, AdamW) on one task batch; steps cycle
vqa -> gqa -> nlvr -> caption with the reference's per-task batch sizes (500 / 833 / 166 / 416,
multitask.py:682-695).  ``--scaling weak`` (default): that batch per GPU; ``--scaling strong``: that batch is the
GLOBAL batch, split across the ranks.  Synthetic CLIP-feature + token batches are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.

Other workloads (parity-tested configs, not the headline): ``--model t5`` configs[2] (T5-base, r = 192),
``--model lora`` configs[3] (BART-base + LoRA r = --lora-r on q_proj / v_proj of all 18 attentions: the K3 path),
``--model video`` configs[4] (BART-base + VL-PET-large, video-text: batch 50, 600 text tokens + 64 frame features of 512).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # before torch touches HIP (see vl-pet_amd/__init__.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s measured achievable)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)
# HBM-side bytes per row of the K1 backward from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), collected
# with rocprofv3 --pmc in its own run at ONE size (M = 28,000, bf16, r = 96) -- bench.py scales them by the run's rows per
# launch, so for the other task shapes they are an extrapolation (labelled as such in the line).
# Two tables: the two-pass form (round 3: pass 1 + column-parallel pass + finalize) and the previous split (--k1-previous-split).
PMC_TRAFFIC_FORMS = {
    # round 6: measured at all four task sizes of configs[1] (profiles/r06_pmc_traffic.md; tools/gpu/r6_g.sh); bytes per LAUNCH by rows, interpolated
    # per launch.  r <= 96: pass 2 sums the row-chunk partials itself (no finalize launch), so its figure includes that read.
    "two_pass": {"source": "profiles/r06_pmc_traffic.md", "measured_at_rows": [15272, 28000, 31616, 46648],
                 "bytes_by_rows": {"k1_bwd_rows": {15272: 62.7e6, 28000: 112.8e6, 31616: 127.1e6, 46648: 187.6e6},
                                   "k1_bwd_wgrad": {15272: 226.2e6, 28000: 340.3e6, 31616: 369.4e6, 46648: 496.9e6},
                                   "k1_bwd_op": {15272: 288.9e6, 28000: 453.1e6, 31616: 496.5e6, 46648: 684.5e6}}},
    "two_pass_t5": {"source": "profiles/r06_pmc_traffic.md", "measured_at_rows": [18250, 28000],    # r = 192
                    "bytes_by_rows": {"k1_bwd_rows": {18250: 93.9e6, 28000: 141.5e6}, "k1_bwd_wgrad": {18250: 280.4e6, 28000: 409.2e6},
                                      "k1_bwd_op": {18250: 425.2e6, 28000: 601.6e6}}},
    "previous_split": {"source": "profiles/r02_pmc_traffic_k1_bwd.md", "measured_at_rows": [28000],
                       "bytes_by_rows": {"k1_bwd_rows": {28000: 11881.0 * 28000}, "k1_bwd_op": {28000: 20432.0 * 28000}}},
}


def pmc_bytes(table, rows):
    """PMC-measured bytes of one launch at `rows`: linear interpolation between the measured sizes (the traffic is affine in the rows:
    fixed partial sums + a per-row term), proportional scaling outside them / with a single point."""
    pts = sorted(table.items())
    if len(pts) == 1 or rows <= 0:
        return pts[0][1] * rows / pts[0][0]
    lo = max([p for p in pts if p[0] <= rows] or [pts[0]], key=lambda p: p[0])
    hi = min([p for p in pts if p[0] >= rows] or [pts[-1]], key=lambda p: p[0])
    if lo[0] == hi[0]:
        if rows == lo[0]:
            return lo[1]
        lo, hi = (pts[0], pts[1]) if rows < pts[0][0] else (pts[-2], pts[-1])      # extrapolate along the nearest segment
    return lo[1] + (hi[1] - lo[1]) * (rows - lo[0]) / (hi[0] - lo[0])


def step_time_report(step_ms, step_tasks, batch_of, settling, n_ranks):
    """The per-step distribution of the timed region (rank 0's HIP events on the step boundaries), beside the mean `ms_per_step`: the K
    step times in order, their median / min, the same per task shape (the four shapes differ 5x in rows), the first timed step of each
    shape against that shape's median, and the throughput a run of median steps would give -- so that a reader of ONE driver line can
    tell a fresh-box tail (a few slow steps: steady_state above value) from a slower steady state (both low)."""
    def med(v):
        v = sorted(v)
        n = len(v)
        return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    by_task = {}
    for ms, t in zip(step_ms, step_tasks):
        by_task.setdefault(t, []).append(ms)
    per_task = {t: {"n": len(v), "median": round(med(v), 3), "min": min(v), "max": max(v), "first": v[0],
                    "first_over_median": round(v[0] / med(v), 3)} for t, v in by_task.items()}
    round_ms = sum(per_task[t]["median"] for t in per_task)
    round_samples = sum(batch_of[t] for t in per_task) * n_ranks
    return {"step_ms": step_ms, "step_ms_median": round(med(step_ms), 3), "step_ms_min": min(step_ms), "step_ms_max": max(step_ms),
            "step_ms_by_task": per_task,
            "steady_state": {"value": round(round_samples / round_ms * 1e3, 2), "unit": "samples/s",
                             "note": "samples of one round of task shapes / sum of the per-shape MEDIAN step times (rank 0's events); "
                                     "`value` above stays all samples / wall time of the K timed steps"},
            "slow_steps": [k for k, (ms, t) in enumerate(zip(step_ms, step_tasks)) if ms > 1.15 * per_task[t]["median"]],
            "settling_rounds_ms": settling}


IMAGE_TASKS = ["vqa", "gqa", "nlvr", "caption"]
VIDEO_TASKS = ["tvqa", "how2qa", "tvc", "yc2c"]


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=18.0, warm=2, batch=4):
    """Reference path restated on the host CPU (kind "port"): the same host model with the PET ops routed to
    oracle/vlpet_oracle.py (plain eager PyTorch in the reference's op order), fp32, every host core.
    (i) full train step of BASELINE.json configs[0] (VQA, batch 4) -> value; (ii) the isolated K1-K4 op chains at
    the configs[1] VQA-step sizes, forward + backward, median of a few iterations."""
    import torch
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    from oracle.host_patch import cpu_reference_ops
    from oracle import vlpet_oracle as O

    ncores = os.cpu_count() or 1
    with cpu_reference_ops():
        torch.manual_seed(1234)
        cfg = HB.vlpet_config()
        model = HB.VLBart(cfg)
        TR.trainable_names(model, cfg)
        model.train()
        tr = TR.Trainer(model, cfg, total_steps=1000)
        gen = torch.Generator().manual_seed(1234)
        b = TR.synthetic_batch("vqa", batch, cfg, "cpu", gen)
        # every host core and, because batch-4 eager ops stop scaling long before a server's core
        # count, 16 / 32 / 64 / 128 threads as well: the fastest is the reported baseline, all are in `thread_sweep`
        sweep = {}
        skipped = {}

        def probe_chain():                               # a config-0-sized K1 chain: milliseconds at a sane thread count
            gp = torch.Generator().manual_seed(3)
            xs = [torch.randn(224, 768, generator=gp) for _ in range(3)]
            ws = [torch.randn(96, 768, generator=gp) * 0.05, torch.zeros(96), torch.randn(768, 96, generator=gp) * 0.05, torch.zeros(768)] * 2
            O.k1_fwd_bwd(xs[0], xs[1], *ws, xs[2], n_heads=4)
            t = time.perf_counter()
            for _ in range(3):
                O.k1_fwd_bwd(xs[0], xs[1], *ws, xs[2], n_heads=4)
            return (time.perf_counter() - t) / 3
        base_probe = base_step = None
        for nt in sorted({min(n, ncores) for n in (16, 32, 64, 128, ncores)}):     # 16 threads first: it bounds what the wider legs may cost
            torch.set_num_threads(nt)
            pr = probe_chain()
            if base_probe is None:
                base_probe = pr
            elif pr > 4.0 * base_probe:                  # oversubscribed: a full step would take minutes (285 s on a 256-thread host)
                skipped[nt] = f"skipped: the config-0 K1 chain is {pr / base_probe:.0f}x slower than at {min(16, ncores)} threads"
                print(f"[bench cpu_baseline] {nt} threads: {skipped[nt]}", file=sys.stderr, flush=True)
                continue
            t0 = time.perf_counter()
            tr.step(b)                                   # warm-up (also the probe: a pathological thread count shows here)
            probe = time.perf_counter() - t0
            if base_step is not None and probe > 5.0 * base_step:
                sweep[nt] = (batch / probe, 1, probe)
                print(f"[bench cpu_baseline] {nt} threads: 1 step in {probe:.1f} s (not repeated)", file=sys.stderr, flush=True)
                continue
            if probe < 2.0:
                for _ in range(warm - 1):
                    tr.step(b)
            t0 = time.perf_counter()
            steps = 0
            while steps < 2 or (time.perf_counter() - t0 < budget_s * 0.2 and steps < 400):
                tr.step(b)
                steps += 1
            dt = time.perf_counter() - t0
            sweep[nt] = (batch * steps / dt, steps, dt)
            if base_step is None:
                base_step = dt / steps
            print(f"[bench cpu_baseline] {nt} threads: {steps} steps in {dt:.1f} s", file=sys.stderr, flush=True)
        best = max(sweep, key=lambda k: sweep[k][0])
        rate, steps, dt = sweep[best]
    del model, tr
    torch.set_num_threads(best)                          # the isolated chains run at the better of the two thread counts

    # isolated chains at the VQA-step size (M = 500 * 56 rows; K4: 500 * 36 visual rows), fp32
    def med(fn, n=3):
        t = time.perf_counter()
        fn()
        first = time.perf_counter() - t
        if first > 5.0:                                  # bounded sample: keep the default bench run within minutes
            return first
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2]
    g = torch.Generator().manual_seed(7)
    M, d, r = 28000, 768, 96
    rn = lambda *s: torch.randn(*s, generator=g) * 0.05
    x1, x2, dy = torch.randn(M, d, generator=g), torch.randn(M, d, generator=g), torch.randn(M, d, generator=g)
    W = [rn(r, d), rn(r), rn(d, r), rn(d), rn(r, d), rn(r), rn(d, r), rn(d)]
    chains = {"threads": best}
    chains["k1_fwd_bwd_s"] = med(lambda: O.k1_fwd_bwd(x1, x2, *W, dy, n_heads=4))

    def k2():
        ps = [w.clone().requires_grad_(True) for w in W[:4]]
        xx, yy = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        O.parallel_adapter(xx, yy, *ps, None).backward(dy)
    chains["k2_fwd_bwd_s"] = med(k2)

    def k3():
        A, B = rn(64, d).requires_grad_(True), rn(d, 64).requires_grad_(True)
        xx = x1.clone().requires_grad_(True)
        O.lora_linear(xx, torch.zeros(d, d), None, A, B, 0.5, None, 0.0).backward(dy)
    chains["k3_r64_fwd_bwd_s"] = med(k3)

    def k4():
        Mv, F = 500 * 36, 2048
        feats, pos = torch.randn(Mv // 36, 36, F, generator=g), torch.zeros(Mv // 36, 36, 4)
        ps = [rn(d, F), rn(d), torch.ones(d), torch.zeros(d), rn(d, 5), rn(d), torch.ones(d), torch.zeros(d), rn(2, d), rn(100, d)]
        ps = [p.requires_grad_(True) for p in ps]
        O.visual_embedding(feats, pos, *ps[:8], ps[8], ps[9]).sum().backward()
    chains["k4_fwd_bwd_s"] = med(k4, n=2)
    print(f"[bench cpu_baseline] isolated chains: {chains}", file=sys.stderr, flush=True)
    chains = {k: round(v, 4) for k, v in chains.items()}
    return dict(value=round(rate, 3), unit="samples/s", cores=best, host_cores=ncores,
                cpu_model=cpu_model_name(), kind="port",
                thread_sweep=dict({str(k): round(v[0], 3) for k, v in sweep.items()}, **{str(k): v for k, v in skipped.items()}),
                sample=f"configs[0]: BART-base VL-PET-large r=96, VQA batch {batch}, S=20+36, fp32, full train step "
                       f"(fwd+bwd+clip+AdamW) through oracle/vlpet_oracle.py on the host CPU, {warm} warm-up + "
                       f"{steps} timed steps ({dt:.1f} s)",
                isolated_chains=dict(chains, note="oracle op chains fwd+bwd at the configs[1] VQA-step size (M=28000, "
                                                  "d=768, r=96; K3 r=64; K4 18000x2048->768), fp32, median seconds"))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build_model(args, dev, dtype):
    import vlpet_amd.host.bart as HB
    import vlpet_amd.train as TR
    if args.model == "t5":
        import vlpet_amd.host.t5 as HT
        cfg = HT.vlt5_config()
        model = HT.VLT5(cfg)
        tasks, label = IMAGE_TASKS, "configs[2]: T5-base + VL-PET-large (r=192, gate scale 0.3) image-text multitask"
        metric = "multitask samples/sec (T5-base, r=192)"
    elif args.model == "lora":
        # scripts/image-text/single_lora.sh:53-55: --use_lora --lora_dim R --use_single_lora; lora_alpha 32 (param.py:197),
        # lora_dropout 0.1 (lora/config.py:5-8); the trainable set is lora_* + every bias + visual_embedding
        cfg = HB.vlpet_config(use_adapter=False, use_encoder_adapter_down_multihead=False,
                              use_encoder_adapter_gating_large_x_lowrank=False,
                              use_decoder_enc_attn_value_parallel_adapter_down_dim=False, unfreeze_encoder_layer_norms=False,
                              use_lora=True, lora_dim=args.lora_r, use_single_lora=True)
        model = HB.VLBart(cfg)
        tasks = IMAGE_TASKS
        label = f"configs[3]: BART-base + LoRA (r={args.lora_r}, alpha 32, dropout 0.1, single LoRA) image-text multitask"
        metric = f"multitask samples/sec (BART-base, LoRA r={args.lora_r})"
    elif args.model == "video":
        cfg = HB.vlpet_config(feat_dim=512, n_boxes=64, tasks=",".join(VIDEO_TASKS))
        model = HB.VLBart(cfg)
        tasks = VIDEO_TASKS
        label = "configs[4]: BART-base + VL-PET-large (r=96) video-text multitask (600 text tokens + 64 frame features of 512)"
        metric = "multitask samples/sec (BART-base, r=96, video-text)"
    else:
        cfg = HB.vlpet_config()
        model = HB.VLBart(cfg)
        tasks, label = IMAGE_TASKS, "configs[1]: BART-base + VL-PET-large (r=96) image-text multitask"
        metric = "multitask samples/sec (BART-base, r=96)"
    TR.trainable_names(model, cfg)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    model.to(dev)
    TR.cast_frozen(model, dtype)
    model.train()
    return model, cfg, tasks, label, metric, n_train


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=None,
                    help="VQA-task batch (other tasks scale like the reference); default 500 (bart, lora), 300 (t5), 50 (video)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: --batch per GPU;  strong: --batch is the global batch, partitioned across the ranks.  Default: strong "
                         "when --gpus > 1 (the reference's task batch is global: multitask.py:682-695, scripts/image-text/VL-PET-large.sh:18), "
                         "weak (= the same thing) at one GPU")
    ap.add_argument("--emulate-ranks", type=int, default=1,
                    help="one GPU, the batch ONE of R strong-scaled ranks would see (global task batch / R): a 1-GPU upper-bound "
                         "estimate of the R-GPU step time before any multi-GPU run (value stays this rank's samples/s)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--buckets", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to exercise the DP path on a 1-GPU box")
    ap.add_argument("--pad-mask", action="store_true", default=True,
                    help="(default) build and apply the reference's default attention mask input_ids.ne(pad) every step "
                         "(src/modeling_bart.py:817-818), although the synthetic rows have no pad tokens")
    ap.add_argument("--no-pad-mask", dest="pad_mask", action="store_false",
                    help="tell the host that the rows carry no padding, so it builds no mask (A/B: about +3 %)")
    ap.add_argument("--k1-previous-split", action="store_true",
                    help="A/B: the round-2 form of the gated K1 backward (row kernel + streaming weight gradients) instead of pass 1 + column-parallel pass")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="K1 weight gradients on a side stream (measured 5 %% SLOWER on one MI355X: the step is GPU-bound)")
    ap.add_argument("--gemm-table", default="on", choices=["on", "off", "tune"],
                    help="TunableOp table for the backbone's library GEMMs: on = use vl-pet_amd/tuning/tunableop_gfx950.csv, "
                         "tune = measure this run's shapes into gpurun_out/tunableop_gfx950_new.csv (slow), off = library defaults")
    ap.add_argument("--kernel-table", default="after", choices=["after", "inline", "off"],
                    help="where the per-kernel table is measured: a separate pass after the timed region (default; the timed "
                         "region brackets only the roofline's op), inline (every launch bracketed inside the timed region), off")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="train.Trainer(graph=True): forward + backward of a step captured per task shape with hipGraph and replayed "
                         "(optimizer and gradient exchange eager).  auto = on wherever the trainer can replay (same box, round 6: the "
                         "headline workload 16.99 ms replayed vs 17.93 ms eager -- the eager step's ~150 host launches are no longer hidden "
                         "behind the kernels).  A replayed step runs no host code, and events recorded inside a captured graph cannot be "
                         "read on this stack (tools/graph_event_probe.py), so the same K-step schedule runs once more eagerly right after "
                         "the timed region with the roofline op bracketed (`eager_region`).  off = the eager step timed, brackets inside it")
    ap.add_argument("--capture-collectives", action="store_true",
                    help="graph mode, --gpus > 1: launch the bucket all-reduces from inside the captured backward (RCCL collectives as graph nodes: "
                         "overlap kept under replay) instead of after the replay.  Tested with a one-rank RCCL communicator only.")
    ap.add_argument("--model", default="bart", choices=["bart", "t5", "lora", "video"])
    ap.add_argument("--lora-r", type=int, default=64, help="LoRA rank for --model lora (BASELINE configs[3]: 8 / 64; script: 128)")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.emulate_ranks > 1 and args.gpus != 1:
        raise SystemExit("bench.py: --emulate-ranks is a one-GPU estimate")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: start the N ranks ourselves (the reference spawns one process per visible GPU itself,
        # multitask.py:872-898), one process per GPU, rendezvous on 127.0.0.1
        import torch
        if args.backend == "nccl" and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible "
                             f"(use --backend gloo to exercise the data-parallel path on fewer GPUs)")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    n_ranks = dist.get_world_size() if world > 1 else 1     # what the process group actually has

    import vlpet_amd.functional as VF
    import vlpet_amd.train as TR
    from vlpet_amd import _lib
    _lib.load()     # fail loudly before anything is timed
    ab_switches = None
    if os.environ.get("VLPET_AB") == "1":     # same-box A/Bs: the host package's switches are attributes, set here from VLPET_* variables
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import ab_switches as _ab
        ab_switches = _ab.apply()
    gemm_table = None
    if args.gemm_table != "off":     # library-GEMM solution selection for the frozen backbone (TunableOp; train.use_tuned_gemms)
        if args.gemm_table == "tune":
            new_table = os.path.join(ROOT, "gpurun_out", "tunableop_gfx950_new.csv")
            os.makedirs(os.path.dirname(new_table), exist_ok=True)
            if not os.path.exists(new_table) and os.path.exists(TR.TUNED_GEMMS):       # start from the committed table
                import shutil
                shutil.copyfile(TR.TUNED_GEMMS, new_table)
            TR.use_tuned_gemms(new_table, tune=True)
            gemm_table = "tuning (gpurun_out/tunableop_gfx950_new.csv)"
        elif TR.use_tuned_gemms():
            gemm_table = os.path.relpath(TR.TUNED_GEMMS, ROOT)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(1234)      # same initial state on every rank: no parameter broadcast needed
    model, cfg, tasks, label, metric, n_train = build_model(args, dev, dtype)
    if args.batch is None:
        args.batch = {"bart": 500, "lora": 500, "t5": 300, "video": 50}[args.model]
    total_steps = max(args.steps + args.warmup, 10) + 8
    tr = TR.Trainer(model, cfg, lr=1e-3, clip=5.0, total_steps=total_steps, world_size=n_ranks, n_buckets=args.buckets,
                    overlap_wgrad=args.overlap_wgrad, capture_collectives=args.capture_collectives)
    # auto: replayed graphs wherever the trainer can replay.  Rounds 3-5 kept the headline workload (configs[1], full one-GPU batch)
    # eager: its NLVR shape faulted at the second replay (torch's sort-based embedding backward for the 36,000 image-order ids; fixed in
    # round 6, visual._order_lookup) and the eager step was GPU-bound anyway.  With the round-6 kernels it no longer is: 16.99 ms
    # replayed vs 17.93 ms eager on one box (profiles/r06_bench_bart_graph_s2.json.log, r06_bench_bart_s2.json.log)
    want_graph = args.graph in ("on", "auto")
    graph_on = bool(want_graph and tr.enable_graph())      # (False for per-task adapters / a side-stream trainer: those stay eager)
    # a shape runs once eagerly, is captured at its second step and replays from then on: two untimed SETUP steps per task in front of
    # the W warm-up steps (which then already replay), so that --warmup / --steps keep their meaning.  Eager: one setup step per task
    # (first-use costs of a fresh process -- code-object loads of the library GEMMs, the GEMM table, allocator growth: the first bench
    # run on a fresh box measured 23.9 k samples/s against 26.7 k for the second with W = 4 alone)
    setup_steps = 2 * len(tasks) if graph_on else len(tasks)

    def rank_batch(task):
        gb = TR.TASK_BATCH[task](args.batch)
        if args.emulate_ranks > 1:                                     # rank 0 of R strong-scaled ranks
            return gb // args.emulate_ranks + (1 if gb % args.emulate_ranks else 0)
        if args.scaling == "weak":
            return gb
        return gb // n_ranks + (1 if rank < gb % n_ranks else 0)       # strong: partition the global task batch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    batches = {t: TR.synthetic_batch(t, rank_batch(t), cfg, dev, gen, no_padding=not args.pad_mask) for t in tasks}
    order = [tasks[i % len(tasks)] for i in range(args.warmup + args.steps)]
    total_steps = max(args.steps + args.warmup + setup_steps, 10) + 8 + 5 * len(tasks) + args.steps     # (+ the settling rounds, + the eager region after a replayed one)
    tr.total, tr.warmup = total_steps, int(total_steps * 0.1)

    for i in range(setup_steps):
        tr.step(batches[tasks[i % len(tasks)]])
    if graph_on:
        # the synthetic batches move INTO the captured steps' input buffers (what a loader's host-to-device copies would target): a replay
        # then reads its inputs where they are, as an eager step does, instead of copying 200 MB of CLIP features device-to-device first
        for t_ in tasks:
            buf = tr.input_buffers(batches[t_])
            if buf is not None:
                batches[t_] = buf
    for i in range(args.warmup):
        tr.step(batches[order[i]])
    torch.cuda.synchronize()

    def timed_round():
        """One untimed round (a step per task shape), each step bracketed by HIP events on the launch stream: ms per task."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(tasks) + 1)]
        for k, t_ in enumerate(tasks):
            evs[k].record()
            tr.step(batches[t_])
        evs[-1].record()
        torch.cuda.synchronize()
        return {t_: round(evs[k].elapsed_time(evs[k + 1]), 3) for k, t_ in enumerate(tasks)}
    # Settling rounds: a fresh box keeps paying first-use costs for a few steps after the W warm-up steps (code-object
    # loads, allocator growth, clocks), and 20 timed steps cannot tell that tail from a slower steady state.  Before the timed region,
    # whole rounds (one step per task shape, untimed) run until a round's steps are within 1.15x of the previous round's, three extra rounds at
    # most; every round's step times are REPORTED (`settling_rounds_ms`), none is part of `value`.
    settling = [timed_round()]
    while len(settling) < 4:
        settling.append(timed_round())
        settled = all(settling[-2][t_] <= 1.15 * settling[-1][t_] for t_ in tasks)
        if world > 1:       # every rank must run the same number of rounds (each step is a collective): one more unless ALL have settled
            flag = torch.tensor([0.0 if settled else 1.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            settled = float(flag.item()) == 0.0
        if settled:
            break
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Live HIP-event brackets inside the timed region: only the launches of the roofline's op (every bracket is two marker
    # packets on the launch stream; bracketing all ~150 launches of a step cost the step a few percent).  The full per-kernel
    # table comes from a separate pass after the timed region (--kernel-table inline restores the old behaviour).
    dom_names = ("k3_bwd", "k3_fwd") if args.model == "lora" else ("k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin")
    VF.K1_BWD_PREVIOUS_SPLIT = bool(args.k1_previous_split)
    VF.TIMER = VF.KernelTimer(None if args.kernel_table == "inline" else dom_names)
    # per-step times: one HIP event on the launch stream at every step boundary (a marker packet each; `value` stays samples / wall)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    samples = 0
    for i in range(args.warmup, args.warmup + args.steps):
        b = batches[order[i]]
        step_ev[i - args.warmup].record()
        tr.step(b)
        samples += b["input_ids"].shape[0]
    step_ev[args.steps].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = [round(step_ev[k].elapsed_time(step_ev[k + 1]), 3) for k in range(args.steps)]
    peak_gb = round(torch.cuda.max_memory_allocated(dev) / 1e9, 3)     # peak of torch's allocator up to the end of the timed region (all task shapes seen)
    timer, VF.TIMER = VF.TIMER, None
    eager_region = None
    if graph_on:
        # a replayed step runs no host code, so nothing was bracketed above (and an event recorded inside a captured graph cannot be
        # read back: tools/graph_event_probe.py, "invalid resource handle").  The SAME K-step schedule runs once more, eagerly, in this
        # process right after the timed region, with the roofline op bracketed on the launch stream exactly as --graph off brackets
        # it: `roofline` comes from these K steps, and their wall time is reported beside `value` as `eager_region`.
        tr.disable_graph()
        for t_ in tasks:        # (one untimed eager step per task: the allocator's eager blocks)
            tr.step(batches[t_])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        timer = VF.TIMER = VF.KernelTimer(None if args.kernel_table == "inline" else dom_names)
        e0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            tr.step(batches[order[i]])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e_dt = time.perf_counter() - e0
        VF.TIMER = None
        eager_region = {"ms_per_step": round(e_dt / args.steps * 1e3, 3), "steps": args.steps,
                        "samples_per_s_this_rank": round(samples / e_dt, 2),
                        "note": "the same K steps, eager launches, right after the timed region; the roofline op's HIP-event brackets are from these steps"}
    table_timer = None
    if args.kernel_table == "after" and rank == 0:
        table_timer = VF.TIMER = VF.KernelTimer()
    if args.kernel_table == "after":               # every rank steps (the gradient exchange is collective); untimed
        for t_ in tasks:
            tr.step(batches[t_])
        torch.cuda.synchronize()
        VF.TIMER = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([samples], device=dev, dtype=torch.float64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        samples = int(s.item())
    # The reference's task batch is GLOBAL (multitask.py:682-695), so with N > 1 ranks the default line is the strong-scaled job (the
    # batch partitioned across the ranks, a few thousand encoder rows per rank).  The other reading -- the same per-GPU batch on every
    # rank (weak) -- is measured on the same ranks right after the timed region and reported beside it (`other_scaling`).
    strong = None
    if n_ranks > 1:
        other = "weak" if args.scaling == "strong" else "strong"

        def other_batch(t):
            gb = TR.TASK_BATCH[t](args.batch)
            return gb if other == "weak" else gb // n_ranks + (1 if rank < gb % n_ranks else 0)
        sb = {t: TR.synthetic_batch(t, other_batch(t), cfg, dev, gen, no_padding=not args.pad_mask) for t in tasks}
        for t_ in tasks:                              # new shapes: one untimed step per task
            tr.step(sb[t_])
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_s = 0
        for i in range(args.steps):
            b = sb[tasks[i % len(tasks)]]
            tr.step(b)
            n_s += b["input_ids"].shape[0]
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        dts = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(dts, op=dist.ReduceOp.MAX)
        ns = torch.tensor([n_s], device=dev, dtype=torch.float64)
        dist.all_reduce(ns, op=dist.ReduceOp.SUM)
        strong = {"value": round(float(ns.item()) / float(dts.item()), 2), "unit": "samples/s", "scaling": other,
                  "ms_per_step": round(float(dts.item()) / args.steps * 1e3, 3), "steps": args.steps,
                  ("per_gpu_task_batch" if other == "weak" else "global_task_batch"): {t: TR.TASK_BATCH[t](args.batch) for t in tasks},
                  "note": f"same ranks, {other} scaling (timed right after the main region)"}

    exchange = None
    if args.emulate_ranks > 1 and dev.type == "cuda":
        own_group = not dist.is_initialized()
        if own_group:           # (a one-rank RCCL group of this process: rendezvous on 127.0.0.1)
            dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=dev)
        # What the estimate R x value leaves out: the gradient exchange.  Measured here: the flat trainable-gradient buffer through the
        # collective library on THIS process's (one-rank) group -- launch + kernel cost of the call the trainer makes after a replay --
        # and, beside it, the bandwidth term of a ring all-reduce over R ranks on one xGMI link per hop (153 GB/s per direction,
        # MI355X: 7 links per GPU; RCCL's multi-ring schedules can only be faster).  Under graph replay the exchange follows the
        # replayed backward (train.FlatGrads.finish), i.e. it is fully exposed: both are ADDED to the emulated step.
        buf = tr.flat.flat
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(buf)
        e1.record(); torch.cuda.synchronize()
        one_rank_us = e0.elapsed_time(e1) / 20 * 1e3
        R, nbytes = args.emulate_ranks, buf.numel() * 4
        ring_us = 2.0 * (R - 1) / R * nbytes / 153e9 * 1e6
        exchange = {"bytes": nbytes, "one_rank_all_reduce_us": round(one_rank_us, 1), "ring_model_us": round(ring_us, 1),
                    "ring_model": f"2 (R - 1) / R x {nbytes} B / 153 GB/s (one xGMI link per hop), R = {R}"}
        tr.flat.flat.zero_()
        if own_group:
            dist.destroy_process_group()

    if rank == 0:
        esz = 2 if dtype == torch.bfloat16 else 4
        d = cfg.d_model
        agg = timer.summary()
        if table_timer is not None:                 # the other launch groups: one step per task after the timed region
            for name, a in table_timer.summary().items():
                agg.setdefault(name, a)
        # algorithmic bytes per row: K1 fwd reads x1, x2, writes y; K1 bwd (the whole op: rows kernel + weight
        # gradients) reads dy, x1, x2, writes dx1, dx2; K2 / K3 fwd read x, y|base, write out; K2 / K3 bwd read dy, x, write dx;
        # K5 fwd reads y, x1, writes out; K5 bwd reads dout, writes dx1, dy (the saved pre-norm sum is extra traffic)
        # Form of the gated K1 backward at this shape (vlpet_adapter_gate_bwd_form): 2 = pass 1 (k1_dz2_kernel / pet_gate_dz_kernel: reads dy, x2,
        # writes only the [M, r] dpre) + the column-parallel pass (k1_cols_kernel: reads dy, x1, x2, writes dx1, dx2 and the
        # weight-gradient partials) + the finalize launch; otherwise rows kernel + weight-gradient kernels as in round 2.
        k1_tiles = 6 if args.model == "t5" else 3
        # ... asked for THIS run's row counts (every K1 launch size the timers saw), not for a fixed size (ADVICE r03 / VERDICT r04 #11)
        k1_rows = sorted(agg["k1_bwd_rows"]["by_rows"]) if "k1_bwd_rows" in agg else []
        k1_forms = {int(m): int(VF._lib.load().vlpet_adapter_gate_bwd_form(int(m), d, k1_tiles, 1 if dtype == torch.bfloat16 else 0)) for m in k1_rows}
        k1_form = max(set(k1_forms.values()), key=list(k1_forms.values()).count) if k1_forms else \
            VF._lib.load().vlpet_adapter_gate_bwd_form(28000, d, k1_tiles, 1 if dtype == torch.bfloat16 else 0)
        two_pass = k1_form == 2 and not args.k1_previous_split
        per_row = {"k1_fwd": 3 * d * esz, "k1_bwd_rows": (2 if two_pass else 5) * d * esz, "k1_bwd_wgrad": (5 if two_pass else 0) * d * esz,
                   "k1_bwd_fin": 0, "k2_fwd": 3 * d * esz,
                   "k2_bwd": 3 * d * esz, "k3_fwd": 3 * d * esz, "k3_bwd": 3 * d * esz, "k5_fwd": 3 * d * esz,
                   "k5_bwd": 3 * d * esz, "k4_ln_bwd": 3 * d * esz, "k4_pos_fwd": d * esz, "k4_pos_bwd": d * esz,     # (R written / dout read)
                   "rms_fwd": 2 * d * esz, "rms_bwd": 3 * d * esz}      # (K4's LayerNorm backward: dout, xhat read, dpre written)
        d_ff = int(getattr(cfg, "encoder_ffn_dim", 0) or getattr(cfg, "d_ff", 0))
        per_row.update({"ffn_act_fwd": 2 * d_ff * esz, "ffn_act_bwd": 3 * d_ff * esz})   # backbone FFN activation + dropout pass
        # backbone attention on the short-sequence kernels: q, k, v read + o written / q, k, v, o, do read + dq, dk, dv written
        per_row.update({"attn_fwd": 4 * d * esz, "attn_bwd": 8 * d * esz})
        V_head = int(cfg.vocab_size)
        per_row.update({"ce_fwd": V_head * esz, "ce_bwd": 2 * V_head * esz})              # LM-head loss: one read / read + write of the logits
        F_in = int(cfg.feat_dim)
        flops_per_row = {"k4_fwd": 2.0 * F_in * d, "k4_wgrad": 2.0 * F_in * d}
        kernels = {}
        for name, a in agg.items():
            by = per_row.get(name, 0) * a["rows"]
            k = dict(launches=a["launches"], avg_us=round(a["total_us"] / a["launches"], 2),
                     total_ms=round(a["total_us"] / 1e3, 3),
                     algorithmic_GBps=round(by / a["total_us"] / 1e3, 1) if by else None)
            if k["algorithmic_GBps"]:
                k["hbm_frac"] = round(k["algorithmic_GBps"] / HBM_PEAK_GBS, 4)
            if name in flops_per_row:
                tf = flops_per_row[name] * a["rows"] / a["total_us"] / 1e6
                k["TFLOPps"] = round(tf, 1)
                k["mfma_frac"] = round(tf / MFMA_PEAK_TFLOPS, 4)
            kernels[name] = k
        # the K1 backward as ONE op (rows kernel + weight-gradient kernels): SURVEY 8d's 5*d*b per row over their summed time
        if "k1_bwd_rows" in agg and "k1_bwd_wgrad" in agg:
            a, w = agg["k1_bwd_rows"], agg["k1_bwd_wgrad"]
            op_us = a["total_us"] + w["total_us"] + (agg["k1_bwd_fin"]["total_us"] if "k1_bwd_fin" in agg else 0.0)
            gbps = 5 * d * esz * a["rows"] / op_us / 1e3
            kernels["k1_bwd_op"] = dict(launches=a["launches"], avg_us=round(op_us / a["launches"], 2),
                                        total_ms=round(op_us / 1e3, 3), algorithmic_GBps=round(gbps, 1),
                                        hbm_frac=round(gbps / HBM_PEAK_GBS, 4),
                                        note=(("pass 1 + column-parallel pass + finalize launch" if "k1_bwd_fin" in agg else
                                               "pass 1 + column-parallel pass (weight gradients reduced in-launch)") + " of one K1 backward, 5*d*b per row" if two_pass
                                              else "rows kernel + weight-gradient kernels of one K1 backward, 5*d*b per row"))
        # dominant HIP kernel of the hot path by time
        if args.model == "lora":
            dom = "k3_bwd" if "k3_bwd" in agg else "k3_fwd"
        else:
            dom = ("k1_bwd_wgrad" if two_pass and "k1_bwd_wgrad" in agg else "k1_bwd_rows") if "k1_bwd_rows" in agg else "k1_fwd"
        a = agg[dom]
        kernel_achieved = per_row[dom] * a["rows"] / a["total_us"] / 1e3     # GB/s of the dominant kernel alone
        tiles = 6 if args.model == "t5" else 3
        lora_drop = args.model == "lora" and float(getattr(cfg, "lora_dropout", 0.0) or 0.0) > 0.0
        kname = {"k1_bwd_rows": (f"k1_dz2_kernel<{tiles}>" if two_pass and tiles <= 3 else
                                 f"pet_gate_dz_kernel<{args.dtype},{tiles}>" if two_pass else
                                 f"pet_gate_bwd2_kernel<{args.dtype},{tiles}>" if tiles <= 3 else f"pet_bwd_kernel<{args.dtype},{tiles},gate>"),
                 "k1_bwd_wgrad": (f"k1_cols_kernel<{tiles}>" if tiles <= 3 else "k1_cols6_kernel<6>") +
                                 " (column-parallel pass of the K1 backward: reads dy, x1, x2, writes dx1, dx2)",
                 "k1_fwd": f"pet_gate_fwd_kernel<{args.dtype},{tiles}> / k1_down_kernel + k1_up_kernel (by shape)",
                 "k3_bwd": ("ng_dz_kernel + ng_cols_kernel<drop> + wgrad_finalize_kernel" if lora_drop else
                            "ng_dz_kernel + ng_cols_kernel + wgrad_finalize_kernel") + " (one K3 backward, two-pass form of csrc/pet_cols_ng.hip)",
                 "k3_fwd": f"pet_fwd_kernel<{args.dtype},act_id{',drop' if lora_drop else ''}>"}[dom]
        traffic = None
        PMC_TRAFFIC = PMC_TRAFFIC_FORMS[("two_pass_t5" if args.model == "t5" else "two_pass") if (args.model != "lora" and two_pass)
                                        else "previous_split"]
        def pmc_avg(group):          # average PMC bytes per launch over THIS run's launch sizes
            tab = PMC_TRAFFIC["bytes_by_rows"][group]
            br = a["by_rows"]
            return round(sum(pmc_bytes(tab, m) * c[0] for m, c in br.items()) / max(1, sum(c[0] for c in br.values())))
        if dom in PMC_TRAFFIC["bytes_by_rows"] and args.dtype == "bf16" and args.model in ("bart", "t5", "video"):
            traffic = pmc_avg(dom)
        # Headline fraction = the whole K1 backward OP (SURVEY 8d's 5*d*b per row is the op's byte count; the op is three launches):
        # algorithmic bytes / (pass 1 + column-parallel pass + finalize).  The dominant kernel's own figures stay beside it.
        op = kernels.get("k1_bwd_op") if dom in ("k1_bwd_rows", "k1_bwd_wgrad") else None
        if op is not None:
            achieved, launch_us, what = op["algorithmic_GBps"], op["avg_us"], "K1 backward op = " + \
                (("pass 1 + column-parallel pass + finalize launch" if "k1_bwd_fin" in agg else
                  "pass 1 + column-parallel pass (in-launch reduce-scatter where M >= 8192, finalize launch below)") if two_pass
                 else "rows kernel + weight-gradient kernels")
            algo_row = 5 * d * esz
            if traffic:
                traffic = pmc_avg("k1_bwd_op")
        else:
            achieved, launch_us, what, algo_row = kernel_achieved, a["total_us"] / a["launches"], kname, per_row[dom]
        roof = dict(bound="hbm", kernel=what, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    traffic_source=(f"{PMC_TRAFFIC['source']}: PMC passes at M = {PMC_TRAFFIC['measured_at_rows']}, interpolated per launch "
                                    f"over this run's launch sizes") if traffic else None,
                    avg_launch_us=round(launch_us, 2),
                    avg_rows_per_launch=round(a["rows"] / a["launches"], 1),
                    algorithmic_bytes_per_row=algo_row,
                    dominant_kernel=dict(name=kname, avg_launch_us=round(a["total_us"] / a["launches"], 2),
                                         achieved=round(kernel_achieved, 1), frac=round(kernel_achieved / HBM_PEAK_GBS, 4),
                                         note="the op's algorithmic bytes over this ONE launch's time (the pre-round-4 headline)"))
        if op is not None:
            roof["op_frac"] = op["hbm_frac"]
            roof["op_avg_us"] = op["avg_us"]
        per_task = {t: TR.TASK_BATCH[t](args.batch) for t in tasks}
        enc_rows = {t: rank_batch(t) * (TR.TEXT_LEN[t] + (72 if t == "nlvr" else (64 if t in VIDEO_TASKS else 36))) for t in tasks}
        out = {
            "metric": metric, "value": round(samples / dt, 2), "unit": "samples/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": label + ", full train step (fwd+bwd+grad exchange+clip+AdamW), random-init weights",
                       ("per_gpu_task_batch" if (args.scaling == "weak" and args.emulate_ranks == 1) else "global_task_batch"): per_task,
                       "enc_rows_per_step_rank0": enc_rows, "trainable_params": n_train, "parallelism": f"dp{n_ranks}",
                       "backend": args.backend if n_ranks > 1 else None},
            **step_time_report(step_ms, [order[i] for i in range(args.warmup, args.warmup + args.steps)],
                               {t: rank_batch(t) for t in tasks}, settling, n_ranks),
            "peak_memory_GB": peak_gb,
            **({"eager_region": eager_region} if eager_region else {}),
            "roofline": roof, "kernels": kernels, "backbone_gemm_table": gemm_table,
            **({"ab_switches": ab_switches} if ab_switches else {}),
            **({"other_scaling": strong} if strong is not None else {}),
            **({"emulated_ranks": {"ranks": args.emulate_ranks, "estimate_samples_per_s_all_ranks": round(samples / dt * args.emulate_ranks, 2),
                                   **({"gradient_exchange": exchange,
                                       "estimate_with_exchange_samples_per_s_all_ranks": round(
                                           samples * args.emulate_ranks / (dt + args.steps * 1e-6 * (exchange["one_rank_all_reduce_us"] + exchange["ring_model_us"])), 2)}
                                      if exchange else {}),
                                   "note": "one GPU running the batch rank 0 of R strong-scaled ranks would see; value is this one rank's "
                                           "throughput; estimate = R x value ignores the gradient exchange, estimate_with_exchange adds the "
                                           "measured one-rank collective call and the ring bandwidth term to every step (exposed: the "
                                           "exchange follows the replayed backward)"}}
               if args.emulate_ranks > 1 else {}),
            "step_mode": (f"hipGraph replay: forward + loss + backward captured once per task shape (train.Trainer(graph=True); {setup_steps} "
                          "untimed setup steps before the warm-up; the batches live in the captured steps' input buffers, Trainer.input_buffers), "
                          "gradient exchange + clip + AdamW eager; roofline brackets from the "
                          "same K steps run eagerly right after the timed region (eager_region)") if graph_on
                         else f"eager launches (roofline op bracketed inside the timed region; {setup_steps} untimed setup steps, one per task shape, before the warm-up)",
            "attention_mask": ("default input_ids.ne(pad) mask built and applied every step, as the reference does" if args.pad_mask
                               else "none built (--no-pad-mask: the synthetic rows carry no padding)"),
            "kernel_table": {"after": ("roofline op bracketed in the eager region (the timed region's K steps again, eager); " if graph_on
                                       else "roofline op bracketed inside the timed region; ") + "the other launch groups in one step per task after it",
                             "inline": "every launch group bracketed inside the " + ("eager region" if graph_on else "timed region"),
                             "off": "roofline op only"}[args.kernel_table],
        }
        if n_ranks == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
