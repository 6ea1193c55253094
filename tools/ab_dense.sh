B="python bench.py --no-cpu-baseline --ess-window 0 --logl dense --steps 40 --warmup 10"
for v in ${VARIANTS:-base dn base dn}; do
  if [ $v = base ]; then L=ptmcmcsampler_amd/libptmi.so; else L=ab/libptmi_$v.so; fi
  PTMI_LIB=$L $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s %.4g upd/s  launch %.3f ms  step %.3f ms' % ('$v', j['value'], j['roofline']['avg_launch_ms'], j['ms_per_step']))"
done
