#!/bin/bash
# rocprofv3 passes over the callback bench (the split path, csrc/ptmi_split.hip): per-kernel times of the TIMED REGION, then the HBM
# counters in passes of their own.  gpurun --timeout 1500 -- 'CB_ARGS="--callback-kind hip" CB_TAG=callback bash tools/r6_callback.sh; CB_ARGS="--callback-kind norm" CB_TAG=callbacktorch bash tools/r6_callback.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r06cb
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--callback --steps 20 --warmup 5 --no-cpu-baseline --ess-window 0 --also off ${CB_ARGS:-}"
run() {  # name, window, rocprof args...
    local name=$1 win=$2; shift 2
    timeout 500 rocprofv3 --kernel-trace "$@" -d $OUT/$name -o $name -- python $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1
    echo "$name rc=$?"
    python $ROOT/tools/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name.txt --window "$win" > /dev/null
    rm -rf $OUT/$name
}
WIN="20:split_rows_kernel<false, true"
T=${CB_TAG:-callback}
run ${T}_stats "$WIN" --stats
run ${T}_fetch "$WIN" --pmc FETCH_SIZE
run ${T}_write "$WIN" --pmc WRITE_SIZE
grep -h '"metric"' $OUT/${T}_stats.log | cut -c1-400
head -12 $OUT/${T}_stats.txt
