#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel trace of bench.py (tools/gpu/r5_m.sh writes the reduced CSV: start_ns,dur_ns,grid,wg,name).
Steps are cut at the fused AdamW launch; a replayed (hipGraph) step is one whose span is about its busy time.  Prints, per model, the
kernels of an average replayed step by category: launches, busy time, average duration -- and the durations of a few kernels by grid.
usage: trace_steps.py trace.csv [...]"""
import collections, csv, re, sys


def short(n):
    n = n.replace("DF16b", "bf16")
    n = re.sub(r"(Custom_)?Cijk_.*", "library GEMM (hipBLASLt)", n)
    n = re.sub(r"void at::native::|at::native::", "at::", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"^_Z\d+([a-z0-9_]+?)I.*", r"\1", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"[<(].*", "", n)
    return n[:56]


for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["d"] = int(r["start_ns"]), int(r["dur_ns"])
    opt = [i for i, r in enumerate(rows) if "adamw" in r["name"]]
    steps = []
    for k in range(1, len(opt)):
        seg = rows[opt[k - 1] + 1:opt[k] + 1]
        span = seg[-1]["s"] + seg[-1]["d"] - seg[0]["s"]
        busy = sum(r["d"] for r in seg)
        steps.append((seg, span, busy))
    replay = [s for s in steps if s[1] < 1.35 * s[2]]
    print(f"## {path}: {len(steps)} steps in the trace, {len(replay)} replayed (span < 1.35 x busy)")
    if not replay:
        continue
    n = len(replay)
    agg = collections.defaultdict(lambda: [0, 0])
    for seg, _, _ in replay:
        for r in seg:
            a = agg[short(r["name"])]
            a[0] += 1; a[1] += r["d"]
    tot = sum(v[1] for v in agg.values())
    print(f"average replayed step: {sum(len(s[0]) for s in replay) / n:.0f} kernels, busy {tot / n / 1e6:.3f} ms, span {sum(s[1] for s in replay) / n / 1e6:.3f} ms\n")
    print("| kernel | launches / step | us / step | avg us | share |\n|---|---:|---:|---:|---:|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"| {k} | {v[0] / n:.1f} | {v[1] / n / 1e3:.1f} | {v[1] / v[0] / 1e3:.1f} | {100 * v[1] / tot:.1f} % |")
    floor = sorted(r["d"] for seg, _, _ in replay for r in seg)
    print(f"\nshortest kernels of the replayed steps: 1st percentile {floor[len(floor) // 100] / 1e3:.1f} us, 10th {floor[len(floor) // 10] / 1e3:.1f} us, median {floor[len(floor) // 2] / 1e3:.1f} us\n")
    for key in ("tail_fwd", "tail_bwd", "k1_dz2", "k1_dz6", "k1_cols", "wgrad_finalize", "act_dropout"):
        g = collections.defaultdict(list)
        for seg, _, _ in replay:
            for r in seg:
                if key in r["name"]:
                    g[int(r["grid"]) // int(r["wg"])].append(r["d"] / 1e3)
        if g:
            print(f"{key}: workgroups -> avg us: " + ", ".join(f"{w}: {sum(v) / len(v):.1f}" for w, v in sorted(g.items())))
    print()
