#!/bin/bash
# One round's profile of bench.py (eager launches, headline workload only):
#   kernel times        rocprofv3 --kernel-trace --stats
#   HBM traffic         rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE   (separate passes: 3 + 2 TCC slots of 4)
#   SQ counters         two --pmc passes of 8 SQ counters (+ GRBM_GUI_ACTIVE)
# Counter passes carry --kernel-trace only (gpurun refuses --pmc together with the sys / hip / hsa trace domains).
#   bash tools/profile_round.sh r2     (run on the GPU box; then python tools/summarize_profile.py r2 here)
#   bash tools/profile_round.sh r2nb --notebook      the same for the notebook-arithmetic workload (K1nt + K2)
TAG=${1:-r2}
EXTRA=${2:-}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
BENCH="python3 bench.py $EXTRA --no-cpu-baseline --no-other-workloads --no-graph --steps 9 --warmup 3 --full-record /tmp/vbq_profiled_bench_full.json"
mkdir -p $OUT
python3 bench.py $EXTRA --steps 20 --warmup 5 --full-record $OUT/${TAG}_bench_full.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_stats -o ${TAG} -- $BENCH > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o ${TAG} -- $BENCH > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o ${TAG} -- $BENCH > $OUT/${TAG}_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  -d $OUT/${TAG}_sq1 -o ${TAG} -- $BENCH > $OUT/${TAG}_sq1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d $OUT/${TAG}_sq2 -o ${TAG} -- $BENCH > $OUT/${TAG}_sq2.log 2>&1
# every workload of the default run (the API methods, the per-image call, the facade forms, the other BASELINE configs): kernel times only
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_all_stats -o ${TAG}_all -- python3 bench.py $EXTRA --no-cpu-baseline --no-graph --steps 5 --warmup 2 \
  --full-record /tmp/vbq_profiled_bench_all.json > $OUT/${TAG}_all_stats.log 2>&1
ls $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_sq1 | head -30
tail -2 $OUT/${TAG}_sq1.log
