#!/bin/bash
# kernel trace of the callback path with the default SCAM / AM / DE cycle (timed window): gpurun -- 'bash tools/r6_cbmix.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r06cb
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
name=callbackmix_stats
timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/$name -o $name -- python $ROOT/bench.py --callback --callback-kind hip --mix default --steps 10 --warmup 105 --no-cpu-baseline --ess-window 0 --also off > $OUT/$name.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name.txt --window "10:split_rows_kernel<false, true" > /dev/null
rm -rf $OUT/$name
head -16 $OUT/$name.txt | cut -c1-180
