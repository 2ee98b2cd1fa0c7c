"""Register-spill gate: every kernel of libvlpet_hip.so is compiled with -Rpass-analysis=kernel-resource-usage
(csrc/Makefile leaves one .res report per object under vl-pet_amd/build/), and no kernel that a BASELINE configuration dispatches by
default may spill vector registers: a spill inside a loop that also carries counted vmcnt prefetches costs a memory latency per
reload.  Kernels that still spill are listed here BY NAME with the reason they are off the default
path; a new spill anywhere else fails the CPU suite."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "vl-pet_amd", "build")

# (regex on the demangled name, why it is allowed to spill)
ALLOWED = [
    (r"^void pet_bwd_kernel<.*, 6, ", "r = 192 recompute-form row kernel: superseded by pet_dz6 + pet_cols6 (saved activations); fp32 = parity mode"),
    (r"^void wgrad_kernel<.*, 6>", "register-tile weight gradients at six tiles: superseded by the streaming / column-parallel kernels"),
    (r"^void wgrad_stream_kernel<6, ", "six-tile streaming weight gradients: only the recompute / explicit-mask forms at r > 96 reach it (the training form runs pet_cols_ng)"),
    (r"^void pet_fwd_kernel<.*, true, (true|false), false, false, 4, 12>", "single-wave gate forward: VLPET_DBG = 64 of a debug build only (default: pet_gate_fwd / pet_fwd2p)"),
    (r"^void pet_gate_bwd2_kernel<float, ", "fp32 IO = parity mode"),
    (r"^void pet_gate_cols2?_kernel<", "round-2 two-pass backward (ABI phases bit 2 / debug switches): the default is pet_dz2 / pet_dz6 + pet_cols / pet_cols6"),
    (r"^void visproj_fwd_kernel<", "4-wave K4 forward: VLPET_K4_WAVES4 of a debug build only"),
    (r"^void visproj_fwd2_kernel<float, ", "fp32 IO = parity mode"),
    (r"^void visproj_fwd2_kernel<__bf16, 24>", "round-2 fused K4 forward: superseded by visproj_gemm_kernel (round 5) wherever d_model is a multiple of 256 (both backbones); left for other widths and for visproj.K4_FORM = \"fused\" A/Bs"),
    (r"^void attn_bwd_kernel<3, 4, true>", "backbone pass (SURVEY 8d: ungraded): three waves per SIMD with 7 spilled registers measured faster than two without (DESIGN.md, round 2 fourth session)"),
]


def _reports():
    files = sorted(glob.glob(os.path.join(BUILD, "*.res")))
    if not files:           # a tree that was never built here: build it (hipcc cross-compiles without a GPU)
        subprocess.run(["make", "-C", os.path.join(ROOT, "vl-pet_amd", "csrc"), "-j", "6"], check=True, capture_output=True)
        files = sorted(glob.glob(os.path.join(BUILD, "*.res")))
    rows = []
    for f in files:
        cur = None
        for line in open(f, errors="replace"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"file": os.path.basename(f), "name": m.group(1), "spill": 0, "scratch": 0, "vgpr": 0, "agpr": 0}
                rows.append(cur)
                continue
            if cur is None:
                continue
            for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                             ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
                m = re.search(pat, line)
                if m:
                    cur[key] = int(m.group(1))
    # binutils' c++filt does not know DF16b (__bf16): demangle it as Dh (also a builtin, so no substitution index moves) and rename
    names = subprocess.run(["c++filt"], input="\n".join(r["name"].replace("DF16b", "Dh") for r in rows), capture_output=True,
                           text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["demangled"] = n.replace("(anonymous namespace)::", "").replace("half", "__bf16")
    return rows


def test_every_object_has_a_resource_report_and_kernels():
    rows = _reports()
    objs = {r["file"] for r in rows}
    assert len(rows) > 300 and {"tail.res", "pet_cols.res", "pet_dz2.res", "visproj.res", "pet_fwd2p.res"} <= objs, sorted(objs)


def test_no_vector_register_spills_on_the_default_path():
    rows = _reports()
    bad, used = [], set()
    for r in rows:
        if r["spill"] == 0:
            continue
        hit = next((i for i, (pat, _) in enumerate(ALLOWED) if re.search(pat, r["demangled"])), None)
        if hit is None:
            bad.append((r["file"], r["demangled"], r["vgpr"], r["agpr"], r["spill"]))
        else:
            used.add(hit)
    assert not bad, "kernels that spill vector registers and are not on the allow-list:\n" + "\n".join(map(str, bad))
    stale = [ALLOWED[i][0] for i in range(len(ALLOWED)) if i not in used]
    assert not stale, f"allow-list entries that no longer match a spilling kernel (remove them): {stale}"


def test_k5_row_kernels_keep_four_waves_per_simd_at_d768():
    """the bf16 d = 768 instantiations of the sublayer-tail kernels (three 8-byte pieces per lane) at <= 128 registers
    (four waves per SIMD) -- and, round 5, their branch-free FULL copies (the ones a d = 768 launch takes): no spills, <= 168 registers
    (three waves per SIMD: these keep TWO rows in flight per wave -- the prefetched row really is prefetched, tools/isa_waits.py -- where
    the branchy copies awaited the next row before reducing the current one)."""
    pat = r"tail_(fwd|bwd)_kernel<__bf16, 3, (true|false), (true|false), 8(, (true|false))?(, (true|false))?>"
    rows = [r for r in _reports() if r["file"] == "tail.res" and re.search(pat, r["demangled"])]
    assert len(rows) >= 12, [r["demangled"] for r in rows]
    for r in rows:
        args = re.search(r"<(.*)>\(TailArgs\)", r["demangled"]).group(1).split(", ")
        full = len(args) == 7 and args[6] == "true"
        dres = r["demangled"].startswith("void tail_bwd") and len(args) >= 6 and args[5] == "true"      # (T5's parked-gradient form: a fourth row stream)
        assert r["spill"] == 0, r
        assert r["vgpr"] + r["agpr"] <= (168 if (full or dres) else 128), r
