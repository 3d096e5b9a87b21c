#!/bin/bash
# usage: bash tools/ab_run.sh "<what>" tag1 tag2 ...   (on the GPU box; writes gpurun_out/ab_<first tag>.txt)
WHAT=$1; shift
OUT=gpurun_out/ab_$1.txt
mkdir -p gpurun_out
: > $OUT
for t in "$@"; do
  VBQ_HIP_LIBRARY=$PWD/tools/bin/libvbq_$t.so timeout 600 python3 tools/abtime.py --what $WHAT ${AB_CHECK:+--check} >> $OUT 2>&1
done
cat $OUT
