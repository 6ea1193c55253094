"""Step time of the fused kernels with parameter groups (PTMCMCSampler.py:129-145) against the same cycle without (developer tool, one GPU).
usage: groups_timing.py [ndim] [nwalkers] [--per-walker]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmcsampler_amd.engine import PTEngine

args = [v for v in sys.argv[1:] if not v.startswith("--")]
d = int(args[0]) if len(args) > 0 else 100
W = int(args[1]) if len(args) > 1 else 4096
COV = "per_walker" if "--per-walker" in sys.argv else "pooled"
nt = 64
half = d // 2
for name, groups, weights in (("scam, one group", None, (20, 0, 0)), ("scam, 3 groups", [np.arange(d), np.arange(half), np.arange(half, d)], (20, 0, 0)),
                              ("default mix, one group", None, (20, 20, 20)), ("default mix, 3 groups", [np.arange(d), np.arange(half), np.arange(half, d)], (20, 20, 20))):
    e = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=weights, cov_update=1000, burn=200, tskip=100, seed=1, cov_mode=COV, groups=groups,
                 eig_mode="ql" if COV == "per_walker" else "lapack")
    e.init_state(np.zeros(d))
    e.run(400)
    e.sync()
    t = time.perf_counter()
    e.run(600)
    e.sync()
    dt = time.perf_counter() - t
    print("%-26s %8.3f ms per 100 iterations  %.3g updates/s  variant %s" % (name, dt / 6 * 1e3, nt * W * 600 / dt, e.last_variant()), flush=True)
    del e
