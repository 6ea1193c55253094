"""The reference's own NUTS workload (tests/test_nuts.py:173-221: 40-d interval-transformed Gaussian, SCAM = AM = DE = NUTS = HMC = 10, HMCsteps = 100,
HMCstepsize = 0.4) as the device family ("interval", a, b), batched: step time on one MI355X (developer tool).  usage: nuts40_timing.py [ntemps] [nwalkers]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmcsampler_amd.engine import PTEngine

args = [v for v in sys.argv[1:] if not v.startswith("--")]
nt = int(args[0]) if len(args) > 0 else 1
W = int(args[1]) if len(args) > 1 else 16384
d, a, b = 40, 0.0, 10.0
pmax, hess = -2.3, 0.25          # near the maximum of one coordinate's density; the test takes the covariance from the Hessian there
cov = np.eye(d) / hess
if "--full" in sys.argv:             # a covariance with off-diagonal terms: the whitening tables are full (the whole-wave layout reads them from global memory)
    A = np.random.default_rng(1).standard_normal((d, d))
    cov = cov + 0.05 * (A @ A.T) / d
for name, gw, hmc in (("SCAM/AM/DE", (0, 0), (0.4, 2, 100)), ("+ NUTS", (10, 0), (0.4, 2, 100)), ("+ NUTS + HMC(<=100 steps)", (10, 10), (0.4, 2, 100)),
                      ("+ NUTS + HMC(<=10 steps)", (10, 10), (0.4, 2, 10))):
    e = PTEngine(d, nt, W, cov, logl=("interval", np.full(d, a), np.full(d, b)), logp=("flat",), weights=(10, 10, 10), grad_weights=gw, hmc=hmc,
                 cov_update=500, burn=500, tskip=100, seed=1, cov_mode="pooled")
    e.init_state(np.full(d, pmax))
    e.run(600)
    e.sync()
    t = time.perf_counter()
    e.run(400)
    e.sync()
    dt = time.perf_counter() - t
    gj = e.get("gj") if e.t.get("gj") is not None else None
    print("%-28s %9.3f ms per 100 iterations  %.3g updates/s  leapfrogs per iteration and chain %s" % (
        name, dt / 4 * 1e3, nt * W * 400 / dt, "-" if gj is None else "%.2f" % (gj[..., 7].sum() / (nt * W * 1000.0))), flush=True)
    del e
