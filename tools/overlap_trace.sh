#!/bin/bash
# Kernel trace of one overlapped configuration (K2 of chunk j on a second stream under K1 of chunk j+1): evidence for the
# negative result of profiles/r2_overlap_sweep.txt.   bash tools/overlap_trace.sh   (GPU box)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rocprofv3 --output-format csv --kernel-trace -d $OUT/r2_overlap -o ov -- python3 bench.py --no-cpu-baseline --no-other-workloads --no-graph --chunks 3 --steps 6 --warmup 3 > $OUT/r2_overlap.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.path.join(os.environ.get("PWD"), "gpurun_out", "r2_overlap", "*kernel_trace.csv")) + glob.glob(os.path.join(os.environ.get("PWD"), "gpurun_out", "r2_overlap", "*", "*kernel_trace.csv")))[-1]
rows = [r for r in csv.DictReader(open(f)) if "k_quant_fast<10, 0>" in r["Kernel_Name"] or "k_hist_flat" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full step: last 3 K1 launches and the K2 launches around them
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
k1 = [r for r in rows if "k_quant_fast" in r["Kernel_Name"] and dur(r) < 300.0]     # the chunked launches of the timed steps
k1 = k1[-6:-3]                                                                       # one whole step, not the last one
t0 = int(k1[0]["Start_Timestamp"])
sel = [r for r in rows if t0 - 1000 <= int(r["Start_Timestamp"]) <= t0 + 900_000][:6]
out = ["K1 / K2 of one timed step with 3 row chunks, K2 on a second stream (rocprofv3 --kernel-trace; microseconds from the first K1 launch):", ""]
for r in sel:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = "K1 k_quant_fast" if "k_quant_fast" in r["Kernel_Name"] else "K2 k_hist_flat "
    out.append(f"  {name}  start {s:8.1f}  end {e:8.1f}  duration {e - s:7.1f}   stream/queue {r.get('Queue_Id', '?')}")
out += ["", "Back to back (one launch each, profiles/r2_kernel_stats.csv): K1 ~395-411 us, K2 ~165 us, i.e. ~132 + 55 us per third.",
        "Overlapped, each third of K1 takes as long as K1 + K2 of a third would take one after the other: the co-resident K2 waves only take issue slots from K1."]
open(os.path.join(os.environ.get("PWD"), "gpurun_out", "r2_overlap_trace.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
