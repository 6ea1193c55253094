# GPU tests + the six kernel timings watched while tuning (developer tool): gpurun --timeout 1500 -- 'bash tools/quick_check.sh'
# prints updates/s and the average MH launch (100 steps) in ms per workload
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -3
B="python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '%.4g'%j['value'], j['roofline']['avg_launch_ms'])"; }
run scam --steps 100 --warmup 20
run am_only --weights 0,20,0 --steps 30 --warmup 10
run mix_chain --mix default --steps 60 --warmup 110
run mix_walker --mix default --pick walker --steps 60 --warmup 110
run dense --logl dense --steps 40 --warmup 10
run c5 --logl curved --ndim 20 --ntemps 16 --mix nuts --steps 6 --warmup 4
