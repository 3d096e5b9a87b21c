#!/bin/bash
# per seed as round 3: stress_parity.py --n 8000000 --rounds 40; --levels --n 8000000 --rounds 18; --small --n 2000000 --rounds 6; stress_notebook.py 1000000 10 seed
# usage (GPU box): bash tools/soak_round.sh <tag> <first seed> <n seeds>   -> gpurun_out/<tag>_soak.txt (last line of every log)
TAG=${1:-r4}; S0=${2:-71}; NS=${3:-3}
OUT=gpurun_out/${TAG}_soak.txt
mkdir -p gpurun_out
echo "# Soak of the round's FINAL build, seed offsets $S0..$((S0+NS-1)); the second block repeats seed $S0 with slots reserved for a" > $OUT
echo "# collective (VBQ_RESERVED_WORKGROUPS=64: shrunken resident grids) and with short-lived K1 workgroups (VBQ_K1_DYNAMIC=1)" >> $OUT
for ((s=S0; s<S0+NS; s++)); do
  echo "seed $s k1: $(timeout 900 python3 tools/stress_parity.py --seed $s --n 8000000 --rounds 40 2>&1 | tail -1)" >> $OUT
  echo "seed $s levels: $(timeout 900 python3 tools/stress_parity.py --levels --seed $s --n 8000000 --rounds 18 2>&1 | tail -1)" >> $OUT
  echo "seed $s small: $(timeout 900 python3 tools/stress_parity.py --small --seed $s --n 2000000 --rounds 6 2>&1 | tail -1)" >> $OUT
  echo "seed $s notebook: $(timeout 900 python3 tools/stress_notebook.py 1000000 10 $s 2>&1 | tail -1)" >> $OUT
done
for env in "VBQ_RESERVED_WORKGROUPS=64" "VBQ_RESERVED_WORKGROUPS=700" "VBQ_K1_DYNAMIC=1"; do
  echo "$env seed $S0 k1: $(env $env timeout 900 python3 tools/stress_parity.py --seed $S0 --n 8000000 --rounds 10 2>&1 | tail -1)" >> $OUT
  echo "$env seed $S0 levels: $(env $env timeout 900 python3 tools/stress_parity.py --levels --seed $S0 --n 8000000 --rounds 6 2>&1 | tail -1)" >> $OUT
done
cat $OUT
