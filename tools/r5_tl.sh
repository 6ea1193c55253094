#!/bin/bash
# device timeline of one covariance period of config 4's share (developer tool): which kernels ran when
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_am_rle_gpu.py -m gpu -q -x -k "side_stream" 2>&1 | tail -4
OUT=$(pwd)/gpurun_out/tl4; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT -o tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --ess-window 0 --also off --ndim 1000 --nwalkers 512 --steps 30 --warmup 20 --stats-async ${1:-off} > $OUT/log.txt 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$OUT/tl_results.db")
rows = c.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall() if "stream_id" in [r[1] for r in c.execute("pragma table_info(kernels)")] else [r + (0, 0) for r in c.execute("select name, start, end from kernels order by start").fetchall()]
mh = [r for r in rows if "mh_steps" in r[0]]
t0 = mh[29][1]
t1 = mh[41][2] if len(mh) > 41 else mh[-1][2]
agg = {}
for name, s, e, st, q in rows:
    if s < t0 or s > t1: continue
    short = name.split("(")[0].replace("void ", "")[:50]
    if (e - s) > 150e3 or "mh_steps" in name:
        print("%9.3f ms  +%8.3f ms  q%s  %s" % ((s - t0) * 1e-6, (e - s) * 1e-6, q, short))
    else:
        a = agg.setdefault(short, [0, 0.0]); a[0] += 1; a[1] += (e - s) * 1e-6
print("window %.3f ms" % ((t1 - t0) * 1e-6))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("   small: %-50s %5d calls %8.3f ms" % (k, n, t))
PY
rm -rf $OUT
