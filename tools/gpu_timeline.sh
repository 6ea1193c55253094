#!/bin/bash
# device timeline of the bench (busy per kernel + idle gaps per step): gpurun --timeout 600 -- 'bash tools/gpu_timeline.sh [bench args]'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/timeline
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT -o tl -- python $ROOT/bench.py --no-cpu-baseline --ess-window 0 --steps 40 --warmup 10 "$@" > $OUT/log.txt 2>&1
python $ROOT/tools/timeline.py $OUT/tl_results.db ${TL_SKIP:-12}
rm -f $OUT/tl_results.db
