#!/bin/bash
# usage (GPU box): bash tools/ab_step.sh tag1 tag2 ...  -> the bench step's stage times with each variant library, three runs each
mkdir -p gpurun_out
for rep in 1 2 3; do
for t in "$@"; do
  VBQ_HIP_LIBRARY=$PWD/tools/bin/libvbq_$t.so python3 bench.py --no-cpu-baseline --no-other-workloads --steps 20 --warmup 5 --full-record /tmp/x.json 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['stages_ms'])"
done
done
