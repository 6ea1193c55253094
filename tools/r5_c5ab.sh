#!/bin/bash
# config-5 share with A/B libraries: tools/r5_c5ab.sh NAME...   (base = the in-tree library)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
A="--logl curved --ndim 20 --ntemps 16 --mix nuts --steps 6 --warmup 4"
for v in base "$@"; do
  if [ $v = base ]; then L=ptmcmcsampler_amd/libptmi.so; else L=ptmcmcsampler_amd/libptmi_$v.so; fi
  PTMI_LIB=$L timeout 200 python -m pytest tests/test_gradjump_gpu.py -m gpu -q -x 2>&1 | tail -1
  PTMI_LIB=$L timeout 100 python tools/gj_leap_timing.py 20 curved 2>&1 | tail -1
  for i in 1 2; do PTMI_LIB=$L timeout 120 python bench.py --no-cpu-baseline --ess-window 0 --also off $A 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v c5 %.4g upd/s  launch %.3f ms' % (j['value'], j['roofline']['avg_launch_ms']))"; done
done
