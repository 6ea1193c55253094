import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from ptmcmcsampler_amd.engine import PTEngine
d, nt, W = 20, 16, 4096
for md in (24, 6, 3, 1):
    e = PTEngine(d, nt, W, np.eye(d), logl=("curved",), logp=("box", np.full(d, -10.0), np.full(d, 10.0)), weights=(10, 0, 10), grad_weights=(10, 0),
                 cov_update=1000, burn=10000, tskip=100, seed=1234, cov_mode="pooled", nuts_maxdepth=md)
    e.init_state(np.array(([-0.1, -0.5] * d)[:d]))
    e.run(400); e.sync()
    t = time.perf_counter(); e.run(600); e.sync(); dt = (time.perf_counter() - t) / 6
    print("maxdepth %2d: %.2f ms per 100 iterations, %.3g updates/s" % (md, dt * 1e3, d and nt * W * 100 / dt))
    del e
