#!/bin/bash
for r in 1 2 4; do
  echo "rounds=$r: $(VBQ_HULL_ROUNDS=$r python bench.py --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stages_ms']['pass1_k1h_solve_and_level_histogram'],4), d['parity_vs_oracle_on_sample'])")"
  echo "rounds=$r 1e8: $(VBQ_HULL_ROUNDS=$r python bench.py --steps 5 --warmup 2 --workload synthetic_1e8 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stages_ms']['pass1_k1h_solve_and_level_histogram'],4), d['parity_vs_oracle_on_sample'])")"
done
