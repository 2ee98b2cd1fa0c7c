#!/usr/bin/env python3
"""Feasibility / upside probe for a hipGraph-captured train step: forward + loss + backward of one task batch captured with
torch.cuda.graph (the optimizer stays eager), replayed against the eager step.  usage: graph_probe.py [--model bart] [--emulate-ranks 8]"""
import argparse, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import vlpet_amd.train as TR
import vlpet_amd.functional as VF

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="bart"); ap.add_argument("--emulate-ranks", type=int, default=8)
ap.add_argument("--lora-r", type=int, default=64); ap.add_argument("--steps", type=int, default=12)
args = ap.parse_args()
dev = torch.device("cuda", 0)
TR.use_tuned_gemms()
torch.manual_seed(1234)
model, cfg, tasks, label, metric, n_train = B.build_model(args, dev, torch.bfloat16)
batch0 = {"bart": 500, "lora": 500, "t5": 300, "video": 50}[args.model]
tr = TR.Trainer(model, cfg, lr=1e-3, clip=5.0, total_steps=200)
gen = torch.Generator(device=dev).manual_seed(1234)
def rb(t):
    gb = TR.TASK_BATCH[t](batch0)
    return gb // args.emulate_ranks + (1 if gb % args.emulate_ranks else 0)
batches = {t: TR.synthetic_batch(t, rb(t), cfg, dev, gen, no_padding=False) for t in tasks}
for i in range(4):
    for t in tasks:
        tr.step(batches[t])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    tr.step(batches[tasks[i % len(tasks)]])
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / args.steps * 1e3:.3f} ms/step (emulate_ranks={args.emulate_ranks}, model={args.model})", flush=True)

graphs = {}
pool = None
for t in tasks:
    b = batches[t]
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            per_token, _ = model(b["input_ids"], b["vis_inputs"], b["labels"], b["task"], attention_mask=b.get("attention_mask"),
                                 no_padding=bool(b.get("no_padding", False)))
            loss = TR.task_loss(per_token, b["labels"], b.get("scores"), b["task"])
            loss.backward()
        pool = g.pool()
        graphs[t] = (g, loss)
        print(f"captured {t}", flush=True)
    except Exception:
        traceback.print_exc()
        print(f"capture of {t} FAILED", flush=True)
        break
if len(graphs) == len(tasks):
    def gstep(t):
        g, loss = graphs[t]
        g.replay()
        tr.flat.finish(average=False)
        tr.optim.step(TR.lr_at(tr.step_idx, tr.base_lr, tr.warmup, tr.total))
        tr.step_idx += 1
        tr.flat.begin_step(zero=False)
        return loss
    for t in tasks:
        gstep(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        l = gstep(tasks[i % len(tasks)])
    torch.cuda.synchronize()
    print(f"graph replay + eager optimizer: {(time.perf_counter() - t0) / args.steps * 1e3:.3f} ms/step, last loss {float(l):.4f}", flush=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        graphs[tasks[i % len(tasks)]][0].replay()
    torch.cuda.synchronize()
    print(f"graph replay only: {(time.perf_counter() - t0) / args.steps * 1e3:.3f} ms/step", flush=True)
