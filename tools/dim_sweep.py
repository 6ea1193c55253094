"""Step time of the default SCAM / AM / DE mix over ndim (pooled covariance, 64 x 4096 chains up to 104-d, fewer beyond): a look for slow paths
(developer tool, one GPU).  usage: dim_sweep.py [--per-walker] [--box] [--dense] [--scam] [--dense-callback]
--dense-callback: the dense Gaussian as a GEMM callback on the split path (PTEngine.dense_logl_callback) instead of the built-in family."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmcsampler_amd.engine import PTEngine

cov_mode = "per_walker" if "--per-walker" in sys.argv else "pooled"
for d in (5, 8, 20, 32, 50, 80, 100, 104, 105, 200, 416, 417, 1000):
    nt, W = 64, (4096 if d <= 104 else (1024 if d <= 416 else 256))
    kw = dict(weights=(20, 0, 0) if "--scam" in sys.argv else (20, 20, 20), cov_update=1000, burn=200, tskip=100, seed=1, cov_mode=cov_mode, eig_mode="ql" if cov_mode == "per_walker" and d <= 128 else "lapack")
    if cov_mode == "per_walker" and d > 128:
        W = min(W, 64)
    if "--box" in sys.argv:
        kw.update(logp=("box", np.full(d, -10.0), np.full(d, 10.0)))
    if "--dense" in sys.argv:
        A = np.random.default_rng(0).standard_normal((d, d))
        kw.update(logl=("dense", np.zeros(d), np.linalg.inv(A @ A.T / d + np.eye(d))))
    if "--dense-callback" in sys.argv:
        A = np.random.default_rng(0).standard_normal((d, d))
        P = np.linalg.inv(A @ A.T / d + np.eye(d))
        e = PTEngine(d, nt, W, np.eye(d) * 0.01, split=True, **kw)
        cb = e.dense_logl_callback(np.zeros(d), P)
        e.init_state_callback(np.zeros(d), cb, None)
        e.run = lambda n, e=e, cb=cb: e.run_callback(n, cb, None)
    else:
        e = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
        e.init_state(np.zeros(d))
    e.run(300)
    e.sync()
    t = time.perf_counter()
    e.run(300)
    e.sync()
    dt = time.perf_counter() - t
    print("ndim %4d  %d x %d chains  %8.3f ms per 100 iterations  %.3g updates/s  %.3g element-updates/s  variant %s" % (
        d, nt, W, dt / 3 * 1e3, nt * W * 300 / dt, nt * W * 300 * d / dt, e.last_variant() if "--dense-callback" not in sys.argv else "split path"), flush=True)
    del e
