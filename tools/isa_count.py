#!/usr/bin/env python3
"""Developer tool: instruction mix of the gfx950 kernels of one .hip source, without a GPU.

    python tools/isa_count.py vbq_amd/csrc/vbq_quantize_fast.hip [kernel-substring] [-D...]

Compiles the source device-only to assembly (VBQ_ONLY_N10), then prints per kernel and per basic block above 60
instructions: instruction count, the estimated issue cycles per wave64 (2 for the full-rate VALU ops measured in
tools/ubench.hip -- v_add/sub/mul_f32, and/or/xor, add/sub_u32, lshrrev/lshlrev; 4 for the other VALU ops; 8 for
transcendentals; 16 for f64 rcp/div helpers), LDS / VMEM / SALU / s_nop counts.  A proxy for A/B decisions that
would otherwise each cost a GPU run; the GPU measurement decides.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_RATE = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
             "v_subrev_u32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_not_b32", "v_ashrrev_i32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}


def cycles(m):
    base = m.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if base in TRANS:
        return 16 if base.endswith("f64") else 8
    if base in FULL_RATE and not m.endswith("_sdwa"):
        return 2
    return 4


def compile_asm(src, defs):
    out = tempfile.mktemp(suffix=".s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-DVBQ_ONLY_N10",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "vbq_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", out] + defs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    return out


def main():
    src = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    pats = [a for a in sys.argv[2:] if not a.startswith("-")]
    asm = src if src.endswith(".s") else compile_asm(src, defs)
    fn = None
    blocks = collections.OrderedDict()
    cur = None
    for line in open(asm):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            fn = m.group(1)
            cur = (fn, "entry")
            blocks[cur] = []
            continue
        if fn is None:
            continue
        if s.startswith(".Lfunc_end"):
            fn = None
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur = (fn, m.group(1))
            blocks[cur] = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[cur].append(s.split()[0])
        if s.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):     # a branch ends the block even without a label
            cur = (cur[0], cur[1] + "+")
            blocks[cur] = []
    per_fn = collections.OrderedDict()
    for (f, b), ins in blocks.items():
        per_fn.setdefault(f, []).append((b, ins))
    for f, bl in per_fn.items():
        dem = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()
        if pats and not any(p in dem for p in pats):
            continue
        allins = [i for _, ins in bl for i in ins]
        print(f"== {dem[:150]}")
        print("   total:", summarize(allins))
        for b, ins in bl:
            if len(ins) >= 60:
                print(f"   {b:14s}", summarize(ins))
                top = collections.Counter(i for i in ins if i.startswith("v_")).most_common(14)
                print("                 ", " ".join(f"{k}:{v}" for k, v in top))


def summarize(ins):
    valu = [i for i in ins if i.startswith("v_")]
    cyc = sum(cycles(i) for i in valu)
    lds = sum(1 for i in ins if i.startswith("ds_"))
    vmem = sum(1 for i in ins if i.startswith(("global_", "buffer_", "flat_", "scratch_")))
    salu = sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_nop", "s_waitcnt")))
    nop = sum(1 for i in ins if i.startswith("s_nop"))
    wait = sum(1 for i in ins if i.startswith("s_waitcnt"))
    return f"{len(ins)} instr, VALU {len(valu)} (~{cyc} cyc), LDS {lds}, VMEM {vmem}, SALU {salu}, s_nop {nop}, waitcnt {wait}"


if __name__ == "__main__":
    main()
