#!/bin/bash
# One rocprofv3 counter pass over a bench command (counters only with --kernel-trace: never with --stats or trace domains):
#   bash tools/pmc_one.sh <name> "<counters>" <bench args...>     -> gpurun_out/<name>.txt
set -u
NAME=$1; PMC=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc $PMC -d $ROOT/gpurun_out/$NAME -o $NAME -- python $ROOT/bench.py --no-cpu-baseline --ess-window 0 "$@" > $ROOT/gpurun_out/$NAME.log 2>&1
python $ROOT/tools/rocpd_summary.py $ROOT/gpurun_out/$NAME/${NAME}_results.db $ROOT/gpurun_out/$NAME.txt > /dev/null
rm -rf $ROOT/gpurun_out/$NAME
grep -E "mh_pc_kernel|mh_steps_kernel|pool_syrk|Counter" $ROOT/gpurun_out/$NAME.txt | grep -v "^void at" | cut -c1-60,86-140 | head -40
