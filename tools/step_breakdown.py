"""Per-step kernel breakdown of a bench.py run traced with `rocprofv3 --kernel-trace --output-format csv`: the K timed (replayed) steps are
found from the bench line (setup + warm-up + settling rounds) and the optimizer launch that ends every step.
usage: python tools/step_breakdown.py kernel_trace.csv bench.json.log [top]"""
import collections
import csv
import json
import sys


def main():
    trace, line = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    j = json.loads([ln for ln in open(line) if ln.startswith("{")][0])
    ntask = len(j["step_ms_by_task"])
    setup = (2 if "replay" in j["step_mode"] else 1) * ntask
    first = setup + j["warmup"] + ntask * len(j["settling_rounds_ms"])
    ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    tot, cnt = collections.Counter(), collections.Counter()
    busy = wall = 0
    K = j["steps"]
    for k in range(first, first + K):
        st = rows[ad[k - 1] + 1:ad[k] + 1]
        wall += int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])
        for r in st:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            n = r["Kernel_Name"]
            if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
                n = "library GEMM"
            tot[n] += d; cnt[n] += 1; busy += d
    print(f"{K} timed steps: wall {wall / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms ({busy / wall * 100:.1f} % busy), {sum(cnt.values()) / K:.0f} launches per step")
    for n, v in tot.most_common(top):
        print(f"{v / busy * 100:5.2f}% {v / K / 1e3:8.1f} us/step {cnt[n] / K:6.1f}/step  {n[:150]}")


if __name__ == "__main__":
    main()
