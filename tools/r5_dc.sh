#!/bin/bash
# the divide-and-conquer solver against the library's: accuracy + time alone, a kernel trace of one factorization, config 4 with either
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== own divide and conquer"; python tools/sytrd_check.py 1000 300 2>&1 | grep -v "^$" | tail -4
echo "== library (PTMI_SYTRD_LIB=1)"; PTMI_SYTRD_LIB=1 python tools/sytrd_check.py 1000 300 2>&1 | grep -v "^$" | tail -4
OUT=$(pwd)/gpurun_out/dc_tr; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o tr -- python $GRAFT_REPO_ROOT/tools/sytrd_check.py 1000 > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/tr_results.db $GRAFT_REPO_ROOT/gpurun_out/r5_dc_stats.txt | head -24 | cut -c1-60,86-150
rm -rf $OUT
cd $GRAFT_REPO_ROOT
for L in 9 10; do
python bench.py --no-cpu-baseline --ess-window 0 --also off --ndim 1000 --nwalkers 512 --steps 40 --warmup 20 --eig-lag $L 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('c4 own dc lag $L', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
PTMI_SYTRD_LIB=1 python bench.py --no-cpu-baseline --ess-window 0 --also off --ndim 1000 --nwalkers 512 --steps 40 --warmup 20 --eig-lag $L 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('c4 library lag $L', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done
