#!/bin/bash
# per-walker mode: bench line + per-kernel times (developer tool): bash tools/pw_prof.sh [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pw1
rocprofv3 --kernel-trace --stats -d /tmp/pw1 -o p1 -- python $ROOT/bench.py --no-cpu-baseline --ess-window 0 --cov-mode per_walker_device --steps 20 --warmup 10 "$@" 2>/dev/null | tail -1 | cut -c90-150
python $ROOT/tools/rocpd_summary.py /tmp/pw1/p1_results.db /tmp/pw1.txt >/dev/null; grep -v Cijk /tmp/pw1.txt | sed -n '3,8p' | cut -c1-40,86-150
