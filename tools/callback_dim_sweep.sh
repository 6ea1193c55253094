#!/bin/bash
# The callback path (split_rows_kernel + a HIP callback) over ndim: updates/s and the share of the HBM peak on 16 d + 32 bytes per update.
# gpurun --timeout 900 -- 'bash tools/callback_dim_sweep.sh'
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "5 4096" "20 4096" "50 4096" "100 4096" "104 4096" "200 2048" "416 1024" "1000 512"; do
  set -- $cfg
  python bench.py --callback --callback-kind hip --ndim $1 --nwalkers $2 --cov-mode pooled --steps 10 --warmup 5 --no-cpu-baseline --ess-window 0 --also off 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('ndim %5d  64 x %5d chains  %.4g upd/s  %.1f us per iteration  HBM frac %.3f on %d B/update  acceptance %.2f' % ($1, $2, o['value'], o['ms_per_step']*10, r['frac'], r['algorithmic_bytes_per_update'], r['acceptance_whole_run']))"
done
