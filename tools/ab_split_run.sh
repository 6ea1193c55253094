#!/bin/bash
# gpurun -- 'bash tools/ab_split_run.sh': the callback bench over every A/B build of the split kernels (tools/ab_split.sh)
cd ${GRAFT_REPO_ROOT:-.}
for so in ptmcmcsampler_amd/csrc/build/ab_*.so; do
  for rep in 1 2; do
    PTMI_LIB=$PWD/$so python bench.py --callback --callback-kind ${CBK:-hip} --steps 20 --warmup 5 --no-cpu-baseline --ess-window 0 --also off 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('%-60s %.4g upd/s  segment %.3f ms' % ('$so', o['value'], r['avg_launch_ms']))"
  done
done
