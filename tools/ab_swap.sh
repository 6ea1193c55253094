# swap epoch timing A/B (developer tool)
python tools/epoch_timing.py 2>&1 | tail -4
python bench.py --no-cpu-baseline --ess-window 0 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scam %.4g upd/s  launch %.3f ms  step %.3f ms' % (j['value'], j['roofline']['avg_launch_ms'], j['ms_per_step']))"
python -m pytest tests/test_gpu_parity.py tests/test_sampler_gpu.py -m gpu -q -x 2>&1 | tail -2
