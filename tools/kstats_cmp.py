"""Compare two rocprofv3 kernel_stats.csv files by kernel category (per step).  usage: kstats_cmp.py A.csv B.csv steps"""
import collections
import csv
import sys

CATS = ['attn_bwd', 'attn_fwd', 'wgrad_stream', 'wgrad_fin', 'tail_fwd', 'tail_bwd', 'act_dropout', 'pet_gate_fwd', 'pet_gate_bwd2',
        'pet_fwd', 'pet_bwd', 'visproj', 'ce_', 'downsample', 'adamw', 'CUDAFunctor_add', 'reduce_kernel', 'copyBuffer', 'fillBuffer',
        'bfloat16tofloat32', 'elementwise', 'wgrad']


def cat(n):
    if 'Cijk' in n:
        return 'gemm'
    for k in CATS:
        if k in n:
            return k
    return 'other'


def load(f, steps):
    t, c = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        t[cat(r['Name'])] += int(r['TotalDurationNs']) / steps / 1e3
        c[cat(r['Name'])] += int(r['Calls']) / steps
    return t, c


a, b, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
(ta, ca), (tb, cb) = load(a, steps), load(b, steps)
print(f"{'category':20s} {'A us/step':>10s} {'calls':>7s} | {'B us/step':>10s} {'calls':>7s} | {'B-A us':>8s}")
for k in sorted(set(ta) | set(tb), key=lambda k: -max(ta[k], tb[k])):
    print(f"{k:20s} {ta[k]:10.1f} {ca[k]:7.1f} | {tb[k]:10.1f} {cb[k]:7.1f} | {tb[k] - ta[k]:8.1f}")
print(f"{'total':20s} {sum(ta.values()):10.1f} {'':7s} | {sum(tb.values()):10.1f} {'':7s} | {sum(tb.values()) - sum(ta.values()):8.1f}")
