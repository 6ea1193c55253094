# A/B of the persistent ULDS kernel (developer tool): gpurun --timeout 900 -- 'bash tools/ab_pers.sh'
B="python bench.py --no-cpu-baseline --ess-window 0"
for p in 0 512 768; do
  PTMI_ULDS_PERS=$p $B --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pers=$p  %.4g upd/s  launch %.3f ms  step %.3f ms' % (j['value'], j['roofline']['avg_launch_ms'], j['ms_per_step']))"
done
python -m pytest tests/test_gpu_bench_kernels.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
