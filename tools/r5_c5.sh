#!/bin/bash
# config-5 share: gradient-jump kernel variants (waves per SIMD, the jump as a call or inlined, tree heights kept in LDS)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gradjump_gpu.py -m gpu -q -x 2>&1 | tail -2
A="--logl curved --ndim 20 --ntemps 16 --mix nuts --steps 6 --warmup 4"
bash tools/ab_run.sh "$A" gj3 gj4
for lv in 4 2; do for v in base gj3 gj4; do
  if [ $v = base ]; then L=ptmcmcsampler_amd/libptmi.so; else L=ptmcmcsampler_amd/libptmi_$v.so; fi
  PTMI_GJ_LDS_DEFAULT=$lv PTMI_LIB=$L python bench.py --no-cpu-baseline --ess-window 0 --also off $A 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lds levels $lv %-6s %.4g upd/s  launch %.3f ms' % ('$v', j['value'], j['roofline']['avg_launch_ms']))"
done; done
