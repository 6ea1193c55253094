#!/bin/bash
# tools/ab_run.sh "<bench args>" name1 name2 ...   (base = the in-tree library); two rounds, interleaved
ARGS=$1; shift
for round in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then L=ptmcmcsampler_amd/libptmi.so; else L=ptmcmcsampler_amd/libptmi_$v.so; fi
  PTMI_LIB=$L python bench.py --no-cpu-baseline --ess-window 0 --also off $ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %.4g upd/s  launch %.4f ms  step %.4f ms' % ('$v', j['value'], j['roofline']['avg_launch_ms'], j['ms_per_step']))"
done; done
