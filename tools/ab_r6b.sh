#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/ab_r6b.txt
: > $OUT
for rep in 1 2; do
for t in "$@"; do
  VBQ_HIP_LIBRARY=$PWD/tools/bin/libvbq_$t.so timeout 600 python3 tools/abtime.py --what ${WHAT:-k1e} >> $OUT 2>&1
done
done
grep -v "^$\|amdgpu.ids" $OUT
