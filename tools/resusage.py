#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print a per-kernel register / scratch / occupancy table."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/tmp/_res.o",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True)
txt = out.stderr
if out.returncode:
    print(txt[-6000:]); sys.exit(1)
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print(f"{name[:90]:90s} vgpr={r.get('vgpr')} agpr={r.get('agpr')} scratch={r.get('scratch')} spill={r.get('spill')} occ={r.get('occ')}")
for l in txt.splitlines():
    if "warning" in l: print(l)
