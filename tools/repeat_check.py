"""which of the runs differ: a pooled sytrd run several times in one process (developer tool).  usage: repeat_check.py [eig_mode] [eig_lag] [ndim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ptmcmcsampler_amd.engine import PTEngine
mode = sys.argv[1] if len(sys.argv) > 1 else "sytrd"
lag = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d = int(sys.argv[3]) if len(sys.argv) > 3 else 300
nt, W, cu = 4, 6, 30
kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=10, seed=8, cov_mode="pooled", eig_mode=mode, eig_lag=lag)
runs = []
for asy in (False, False, False, True) if lag else (False, False, False):
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, stats_async=asy, **kw)
    g.init_state(np.zeros(d))
    snaps = []
    for n in (cu + 10, 2 * cu, 7, 3 * cu):
        g.run(n)
        g.sync()
        snaps.append((g.get("X").copy(), g.get("Ut").copy(), g.get("cov").copy(), g.get("S").copy()))
    runs.append(snaps)
    del g
for i, a in enumerate(runs):
    print("run %d:" % i, " ".join("vs%d:%s" % (j, "".join("=" if all(np.array_equal(x, y) for x, y in zip(sa, sb)) else "X" for sa, sb in zip(a, b))) for j, b in enumerate(runs)))
a, b = runs[0], runs[1]
for k, (sa, sb) in enumerate(zip(a, b)):
    print("snap %d: X %.3g  Ut %.3g  cov %.3g  S %.3g" % ((k,) + tuple(float(np.abs(x - y).max()) for x, y in zip(sa, sb))))
