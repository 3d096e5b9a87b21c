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

KERNELS = ("k_quant_fast", "k_hist_flat", "k_transpose", "k_prepare_penalties", "k_quant_tiled", "k_quant_flat",
           "k_gather", "k_hist_tiled", "k_quant_notebook")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


stats = newest(os.path.join(src, f"{tag}_stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
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
    if "FETCH_SIZE_KiB_per_launch" in e and "WRITE_SIZE_KiB_per_launch" in e:
        e["hbm_read_bytes_corrected"] = 2 * e["FETCH_SIZE_KiB_per_launch"] * 1024
        e["hbm_write_bytes"] = e["WRITE_SIZE_KiB_per_launch"] * 1024
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    out[k] = e
bench = os.path.join(src, f"{tag}_bench.json")
if os.path.exists(bench):
    out["bench_line"] = json.loads(open(bench).read().strip().splitlines()[-1])
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary, tag `{tag}`\n\nCommand: `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline` "
            "(kernel times); `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (traffic).\n\n"
            "| kernel | calls | avg us | min us | % of GPU time | HBM read (2 x FETCH_SIZE) MB | HBM write MB |\n|---|---|---|---|---|---|---|\n")
    for k, e in out.items():
        if k == "bench_line":
            continue
        rd, wr = e.get("hbm_read_bytes_corrected"), e.get("hbm_write_bytes")
        tail = f"{rd / 1e6:.1f} | {wr / 1e6:.1f} |" if rd is not None else "- | - |"
        f.write(f"| {k} | {e['calls']} | {e['avg_us']:.1f} | {e['min_us']:.1f} | {e['pct']:.1f} | {tail}\n")
    if "bench_line" in out:
        f.write("\nBench line of the same build (un-profiled run):\n\n```json\n" + json.dumps(out["bench_line"]) + "\n```\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
