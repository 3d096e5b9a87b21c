#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/ab_r6c.txt
: > $OUT
for rep in 1 2 3; do
for t in base cur; do
  VBQ_HIP_LIBRARY=$PWD/tools/bin/libvbq_$t.so timeout 600 python3 tools/abtime.py --what k1nt,k1e >> $OUT 2>&1
done
done
timeout 1200 python3 -m pytest tests/test_gpu_twopass.py tests/test_gpu_parity.py -x -q -m gpu >> $OUT 2>&1
grep -v "^$\|amdgpu.ids" $OUT
