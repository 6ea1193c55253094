# GPU tests + the kernel timings watched while tuning (developer tool): gpurun --timeout 1500 -- 'bash tools/quick_check.sh [notest]'
# prints updates/s, the average MH launch (100 steps) in ms and the wall per step, per workload
if [ "$1" != "notest" ]; then python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -3; fi
B="python bench.py --no-cpu-baseline --ess-window 0"
run() { name=$1; shift; $B "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s %.4g upd/s  launch %.3f ms  step %.3f ms' % ('$name', j['value'], j['roofline']['avg_launch_ms'], j['ms_per_step']))"; }
run scam --steps 100 --warmup 20
run am_only --weights 0,20,0 --steps 30 --warmup 10
run mix_chain --mix default --steps 60 --warmup 110
run mix_walker --mix default --pick walker --steps 60 --warmup 110
run dense --logl dense --steps 40 --warmup 10
run dense_mix_w --logl dense --mix default --pick walker --steps 30 --warmup 110
run c4 --ndim 1000 --nwalkers 512 --steps 40 --warmup 20
run c4_mix --ndim 1000 --nwalkers 512 --mix default --steps 3 --warmup 1
run c5 --logl curved --ndim 20 --ntemps 16 --mix nuts --steps 6 --warmup 4
