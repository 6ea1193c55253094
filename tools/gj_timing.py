"""Per-iteration cost of the fused kernels with and without gradient jumps (one MI355X)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ptmcmcsampler_amd.engine import PTEngine

d, nt, W = 20, 16, 4096
box = ("box", np.full(d, -10.0), np.full(d, 10.0))
p_curved = np.array([-0.1, -0.5] * (d // 2))
for name, kw, p0 in (
        ("curved  SCAM/DE          ", dict(logl=("curved",), logp=box, weights=(10, 0, 10)), p_curved),
        ("curved  SCAM/DE/NUTS     ", dict(logl=("curved",), logp=box, weights=(10, 0, 10), grad_weights=(10, 0)), p_curved),
        ("curved  SCAM/DE/HMC      ", dict(logl=("curved",), logp=box, weights=(10, 0, 10), grad_weights=(0, 10), hmc=(0.08, 2, 50)), p_curved),
        ("iso     SCAM/DE/NUTS     ", dict(logl=("iso",), weights=(10, 0, 10), grad_weights=(10, 0)), np.zeros(d)),
        ("iso     NUTS only        ", dict(logl=("iso",), weights=(0, 0, 0), grad_weights=(10, 0)), np.zeros(d)),
):
    e = PTEngine(d, nt, W, np.eye(d), cov_update=1000, burn=10000, tskip=100, seed=1, cov_mode="pooled", **kw)
    e.init_state(p0)
    e.run(300)
    e.sync()
    t = time.perf_counter()
    e.run(300)
    e.sync()
    dt = time.perf_counter() - t
    js = e.get("jstat").sum(axis=(0, 1))
    gj = e.get("gj") if e.t.get("gj") is not None else None
    print("%s %8.3f ms/iter  %.3g updates/s   jumps %s  eps median %s" % (
        name, dt / 300 * 1e3, nt * W * 300 / dt, js[:, 0].tolist(), "-" if gj is None else "%.3g" % np.median(gj[..., 0])), flush=True)
    del e
    torch.cuda.empty_cache()
