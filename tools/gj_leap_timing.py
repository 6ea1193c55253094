"""Time of one leapfrog on the critical path: ONE wave (16 chains of the config-5 shape), NUTS in every iteration, so the
launch time divided by the wave's leapfrogs (the chains of a wave take their gradient jumps one after the other) is the
latency of a leapfrog plus its share of the tree bookkeeping.  Usage: python tools/gj_leap_timing.py [ndim] [logl]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmcsampler_amd.engine import PTEngine

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "curved"
nt, W = 16, 1
box = ("box", np.full(d, -10.0), np.full(d, 10.0))
logl = ("curved",) if kind == "curved" else ("iso",)
e = PTEngine(d, nt, W, np.eye(d), logl=logl, logp=box, weights=(1, 0, 0), grad_weights=(1000, 0), cov_update=1000, burn=100000,
             tskip=100, seed=1234, cov_mode="pooled")
e.init_state(np.array(([-0.1, -0.5] * d)[:d]))
e.run(300)
e.sync()
g0 = e.get("gj").copy()
t = time.perf_counter()
n = 300
e.run(n)
e.sync()
dt = time.perf_counter() - t
g1 = e.get("gj")
calls = (g1[..., 4] - g0[..., 4]).sum()
leaps = (g1[..., 7] - g0[..., 7]).sum()
print("ndim %d %s: %d iterations of one wave in %.1f ms; %d NUTS calls, %d leapfrogs (%.1f per call): %.2f us per leapfrog, %.1f us per call"
      % (d, kind, n, dt * 1e3, calls, leaps, leaps / max(calls, 1), dt * 1e6 / max(leaps, 1), dt * 1e6 / max(calls, 1)))
