"""Which torch ops launch the small kernels of one train step (configs[1], one task): torch.profiler, grouped by op name and
input shapes, for the aten ops that end in copy / fill / add / cast kernels.  usage: python tools/opcount.py [task]"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                                   # noqa: E402
import vlpet_amd.train as TR                                         # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "vqa"
dev = torch.device("cuda", 0)
args = types.SimpleNamespace(model="bart", lora_r=64)
torch.manual_seed(1234)
model, cfg, tasks, label, metric, n_train = B.build_model(args, dev, torch.bfloat16)
tr = TR.Trainer(model, cfg, lr=1e-3, clip=5.0, total_steps=50, world_size=1, n_buckets=3)
gen = torch.Generator(device=dev).manual_seed(1234)
batch = TR.synthetic_batch(task, TR.TASK_BATCH[task](500), cfg, dev, gen)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile              # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::_to_copy", "aten::sum", "aten::mul",
        "aten::clone", "aten::contiguous", "aten::cat", "aten::div", "aten::zeros")
rows = {}
for e in prof.events():
    if e.name in want and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if "vl-pet_amd" in s or "bench.py" in s][:2]
        k = (e.name, str(e.input_shapes)[:70], " <- ".join(s.split("vl-pet_amd/")[-1][:60] for s in st))
        c = rows.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += e.device_time_total
for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n:4d} x {t / max(n, 1):7.1f} us = {t:8.1f} us  {k[0]:16s} {k[1]:70s} {k[2]}")
