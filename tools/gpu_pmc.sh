#!/bin/bash
# one rocprofv3 counter pass on the bench; usage: gpu_pmc.sh <tag> "<counters>" [bench args]
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT -o pmc -- python $ROOT/bench.py --steps 600 --warmup 100 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
echo "rc=$?"
python $ROOT/tools/rocpd_summary.py $OUT/pmc_results.db $OUT/summary.txt | grep "mh_steps"
