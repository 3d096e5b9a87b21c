#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one round (gpurun_out/<tag>_{stats,fetch,write}) into the
files committed under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc.json, <tag>_summary.md.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of
a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section), so reads are doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

KERNELS = ("k_prep_planes", "k_gather_latents", "k_level_counts_hull", "k_quant_hull_idx", "k_quant_pruned", "k_quant_fast", "k_transpose_batched", "k_rd_sums", "k_hist_flat", "k_transpose", "k_prepare_penalties", "k_quant_tiled",
           "k_quant_flat", "k_gather", "k_hist_tiled", "k_quant_notebook", "k_lut_lengths", "k_lut_models", "k_np_block_sums")


def short(name):
    """Kernel family; the counting instantiation of the fast kernel (template mode 2) is reported on its own."""
    for k in KERNELS:
        if k in name:
            if k == "k_quant_fast" and ("ELi2EE" in name or ", 2>" in name):
                return "k_quant_fast_count"
            return k
    return None


def newest(pattern):
    """Newest match of the pattern with or without rocprofv3's per-host sub-directory."""
    fs = sorted(glob.glob(pattern) + glob.glob(pattern.replace(os.sep + "*" + os.sep, os.sep)), key=os.path.getmtime)
    return fs[-1:] if fs else []


stats = newest(os.path.join(src, f"{tag}_stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
for f_all in newest(os.path.join(src, f"{tag}_all_stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f_all, os.path.join(dst, f"{tag}_all_workloads_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
pmc = collections.defaultdict(dict)
for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = newest(os.path.join(src, f"{tag}_{sub}", "*", "*_counter_collection.csv"))
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == cname:
            acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc[k][cname + "_KiB_per_launch"] = sum(v) / len(v)
        pmc[k]["launches_" + sub] = len(v)
# SQ counters (two more passes; quad-cycle units, MI355X_MICROARCH.md "rocprofv3 PMC slots")
sq = collections.defaultdict(dict)
for sub in ("sq1", "sq2"):
    f = newest(os.path.join(src, f"{tag}_{sub}", "*", "*_counter_collection.csv"))
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    kt = newest(os.path.join(src, f"{tag}_{sub}", "*", "*_kernel_trace.csv"))
    for r in csv.DictReader(open(kt[0])):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, cs in acc.items():
        for cn, v in cs.items():
            sq[k][cn] = sum(v) / len(v)
        sq[k]["dur_us_" + sub] = sum(dur[k]) / len(dur[k])
out = {}
merged = collections.OrderedDict()
for r in rows:                      # template instantiations of one kernel are reported together
    k = short(r["Name"])
    if not k:
        continue
    m = merged.setdefault(k, {"calls": 0, "total": 0.0, "min": 1e30, "max": 0.0, "pct": 0.0})
    m["calls"] += int(r["Calls"]); m["total"] += float(r["TotalDurationNs"])
    m["min"] = min(m["min"], float(r["MinNs"])); m["max"] = max(m["max"], float(r["MaxNs"]))
    m["pct"] += float(r["Percentage"])
for k, m in merged.items():
    e = {"calls": m["calls"], "avg_us": m["total"] / m["calls"] / 1e3, "min_us": m["min"] / 1e3,
         "max_us": m["max"] / 1e3, "pct": m["pct"]}
    e.update(pmc.get(k, {}))
    if k in sq:
        c = dict(sq[k])
        if "GRBM_GUI_ACTIVE" in c and "dur_us_sq1" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0                             # summed over the 8 XCDs
            c["clock_GHz"] = cyc / c["dur_us_sq1"] / 1e3
            simd_quads = cyc * 1024 / 4.0                                # 256 CUs x 4 SIMDs, quad-cycles
            c["valu_active_over_simd_time"] = c.get("SQ_ACTIVE_INST_VALU", 0) / simd_quads
            # share of the SIMDs' issue time spent issuing VALU work: SQ_ACTIVE_INST_VALU counts a quad-cycle per VALU
            # instruction issued, the 2-cycle ops (add / sub / mul / logic) included, so it over-counts those by 2x;
            # with the measured split of the K1 lambda loop (profiles/*_isa.txt: ~55 % two-cycle, 45 % four-cycle
            # instructions) the issue cycles are ~0.72 quad-cycles per instruction
            c["valu_issue_frac"] = min(1.0, 0.72 * c.get("SQ_ACTIVE_INST_VALU", 0) / simd_quads)
            c["waves_per_simd"] = c.get("SQ_WAVE_CYCLES", 0) / simd_quads
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in c and c.get("SQ_WAVE_CYCLES"):
                    c[n + "_frac_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
            if "SQ_LDS_IDX_ACTIVE" in c:
                c["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        e["sq"] = c
    if "FETCH_SIZE_KiB_per_launch" in e and "WRITE_SIZE_KiB_per_launch" in e:
        e["hbm_read_bytes_corrected"] = 2 * e["FETCH_SIZE_KiB_per_launch"] * 1024
        e["hbm_write_bytes"] = e["WRITE_SIZE_KiB_per_launch"] * 1024
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    out[k] = e
bench = os.path.join(src, f"{tag}_bench_final.json")
if not os.path.exists(bench):
    bench = os.path.join(src, f"{tag}_bench.json")
full = os.path.join(src, f"{tag}_bench_full.json")           # the full record of the same run (the line itself is the compact form)
if os.path.exists(full):
    shutil.copy(full, os.path.join(dst, f"{tag}_bench_full.json"))
if os.path.exists(bench):
    out["bench_line"] = json.loads(open(bench).read().strip().splitlines()[-1])
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary, tag `{tag}`\n\nCommands (`tools/profile_round.sh {tag}` on the GPU box): `rocprofv3 --output-format csv "
            "--kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-workloads --no-graph --steps 9 --warmup 3` (kernel "
            "times; eager launches so that every kernel is a trace record; the extra launches beyond steps + warm-up are the bench's own "
            "parity checks), then the same command with `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and two sets of eight SQ counters in "
            "separate passes (`--kernel-trace` only next to `--pmc`).\n\n"
            "| kernel | calls | avg us | min us | % of GPU time | HBM read (2 x FETCH_SIZE) MB | HBM write MB |\n|---|---|---|---|---|---|---|\n")
    for k, e in out.items():
        if k == "bench_line":
            continue
        rd, wr = e.get("hbm_read_bytes_corrected"), e.get("hbm_write_bytes")
        tail = f"{rd / 1e6:.1f} | {wr / 1e6:.1f} |" if rd is not None else "- | - |"
        f.write(f"| {k} | {e['calls']} | {e['avg_us']:.1f} | {e['min_us']:.1f} | {e['pct']:.1f} | {tail}\n")
    f.write("\nSQ counters (separate `--pmc` passes, averages per launch; SQ_* cycle counters are quad-cycles):\n\n"
            "| kernel | clock GHz | VALU instrs | VALU-active / SIMD time | waves per SIMD | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | LDS busy | LDS bank-conflict cycles |\n"
            "|---|---|---|---|---|---|---|---|---|---|\n")
    for k, e in out.items():
        c = e.get("sq") if isinstance(e, dict) else None
        if not c or "clock_GHz" not in c:
            continue
        f.write(f"| {k} | {c['clock_GHz']:.2f} | {c.get('SQ_INSTS_VALU', 0):.3g} | {c['valu_active_over_simd_time']:.2f} | "
                f"{c['waves_per_simd']:.2f} | {c.get('SQ_WAIT_ANY_frac_of_wave_cycles', 0):.2f} | "
                f"{c.get('SQ_WAIT_INST_ANY_frac_of_wave_cycles', 0):.2f} | {c.get('SQ_ACTIVE_INST_ANY_frac_of_wave_cycles', 0):.2f} | "
                f"{c.get('lds_busy_frac', 0):.2f} | {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g} |\n")
    f.write("\nVALU-active / SIMD time can exceed 1: SQ_ACTIVE_INST_VALU counts a quad-cycle per issued VALU instruction, "
            "while add/sub/mul/logic ops issue in 2 cycles per wave64 on gfx950 (tools/ubench.hip).\n")
    isa = os.path.join(dst, f"{tag}_k1_lambda_loop_isa.txt")
    if os.path.exists(isa):
        f.write(f"\nK1 instruction listing of the per-lambda loop: profiles/{os.path.basename(isa)}.\n")
    if "bench_line" in out:
        f.write("\nBench line of the same build (un-profiled run):\n\n```json\n" + json.dumps(out["bench_line"]) + "\n```\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
