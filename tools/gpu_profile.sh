#!/bin/bash
# rocprofv3 passes for the bench kernel; run on the GPU box through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash tools/gpu_profile.sh r01'
# Writes under gpurun_out/prof_<tag>/ ; the summaries to keep are copied to profiles/ by hand.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --steps 1000 --warmup 200 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ARGS > $OUT/stats.log 2>&1
echo "stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $ARGS > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $ARGS > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o sq -- python $ARGS > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?"
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
