#!/bin/bash
# stats_async: parity tests, then on / off timings of config 2 (driver window, default length) and config 4's share (developer tool)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_am_rle_gpu.py tests/test_eig_jacobi.py -m gpu -q -x 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --ess-window 0 --also off"
run() { name=$1; shift; $B "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%-22s %.4g upd/s  launch %.3f ms  step %.3f ms  kernel share %.3f' % ('$name', j['value'], r['avg_launch_ms'], j['ms_per_step'], r['kernel_time_share_of_wall']))"; }
for rep in 1 2; do
run c2_driver_on --steps 20 --warmup 5
run c2_driver_off --steps 20 --warmup 5 --stats-async off
run c2_default_on
run c2_default_off --stats-async off
run c4_on --ndim 1000 --nwalkers 512 --steps 40 --warmup 20
run c4_off --ndim 1000 --nwalkers 512 --steps 40 --warmup 20 --stats-async off
done 2>&1 | tee gpurun_out/r5_async.txt
