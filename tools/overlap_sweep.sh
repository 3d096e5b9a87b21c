#!/bin/bash
# K1 || K2 overlap experiment: row chunks x persistent-grid size, eager and graph.
for ch in 1 2 3 6; do for wg in 4 5; do
  echo "chunks=$ch wg=$wg: $(VBQ_K1_WG_PER_CU=$wg python bench.py --steps 20 --warmup 5 --chunks $ch --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['config']['launch'], {k:round(v,4) for k,v in d['stages_ms'].items()}, d['parity_vs_oracle_on_sample'])")"
done; done
