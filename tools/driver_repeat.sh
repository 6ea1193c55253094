#!/bin/bash
# The driver's command three times over (+ once from cold clocks): run-to-run spread on one box.
# usage: gpurun --timeout 600 -- 'bash tools/driver_repeat.sh'
cd ${GRAFT_REPO_ROOT:-.}
show() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$1 %.4g upd/s  step %.3f ms  launch %.3f ms (first %.3f, last %.3f)  kernel share %.2f' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['launch_ms_first'], r['launch_ms_last'], r['kernel_time_share_of_wall']))"; }
for i in 1 2 3; do python bench.py --no-cpu-baseline --ess-window 0 --steps 20 --warmup 5 2>/dev/null | show "run $i:"; done
python bench.py --no-cpu-baseline --ess-window 0 --steps 20 --warmup 5 --preheat 0 2>/dev/null | show "cold clocks:"
