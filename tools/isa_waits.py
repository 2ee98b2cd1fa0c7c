#!/usr/bin/env python3
"""Compiler-inserted full drains in the device code: for every kernel of the given hipcc -S files, the `s_waitcnt vmcnt(0)` that are NOT
inside an inline-asm block, split into those in loops and those in straight-line code, and the longest run of "load ... vmcnt(0)" pairs
(a chain of dependent memory round trips: the pattern hipcc emits when it cannot move a load above the previous store).
usage: isa_waits.py file.s [...]   (hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o file.s file.hip)"""
import re, subprocess, sys


def demangle(n):
    try:
        return subprocess.run(["c++filt", n.replace("DF16b", "Dh")], capture_output=True, text=True).stdout.strip().replace("half", "__bf16")[:110]
    except Exception:
        return n


for path in sys.argv[1:]:
    L = open(path).read().split("\n")
    i = 0
    while i < len(L):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", L[i])
        if not m:
            i += 1
            continue
        name, j = m.group(1), i + 1
        inasm = False
        in_loop_labels, waits, chain, best = set(), [], 0, 0
        loop_depth_lines = []
        body = []
        while j < len(L) and "s_endpgm" not in L[j]:
            body.append(L[j]); j += 1
        # loop extents: a backward branch to a label defines [label, branch]
        labels = {}
        for k, l in enumerate(body):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = k
        loops = []
        for k, l in enumerate(body):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] <= k:
                loops.append((labels[mm.group(1)], k))
        nloop = nflat = 0
        last_wait = -99
        for k, l in enumerate(body):
            if "ASMSTART" in l: inasm = True
            if "ASMEND" in l: inasm = False
            if not inasm and re.search(r"s_waitcnt\s+vmcnt\(0\)", l):
                if any(a <= k <= b for a, b in loops): nloop += 1
                else: nflat += 1
                # a run: waits separated by fewer than 16 lines with a load in between
                seg = body[last_wait + 1:k] if last_wait >= 0 else []
                if last_wait >= 0 and k - last_wait < 16 and any(re.search(r"(global|buffer|flat)_load", s) for s in seg): chain += 1
                else: chain = 1
                best = max(best, chain)
                last_wait = k
        if nloop or best >= 4:
            print(f"{path.split('/')[-1]:18s} in loops {nloop:3d}  straight-line {nflat:3d}  longest load/wait chain {best:3d}   {demangle(name)}")
        i = j
