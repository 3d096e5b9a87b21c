#!/bin/bash
# usage (GPU box): bash tools/resident_sweep.sh   -> gpurun_out/r4_resident_vs_collective.txt
OUT=gpurun_out/r4_resident_vs_collective.txt
mkdir -p gpurun_out
: > $OUT
for wl in kodak c1; do
  timeout 300 python3 tools/resident_vs_collective.py --workload $wl >> $OUT 2>&1
  VBQ_K1_DYNAMIC=1 timeout 300 python3 tools/resident_vs_collective.py --workload $wl >> $OUT 2>&1
  for r in 16 64 128; do
    VBQ_RESERVED_WORKGROUPS=$r timeout 300 python3 tools/resident_vs_collective.py --workload $wl >> $OUT 2>&1
  done
  timeout 300 python3 tools/resident_vs_collective.py --workload $wl --threads 256 --k 16,64 >> $OUT 2>&1
done
grep -v amdgpu.ids $OUT
