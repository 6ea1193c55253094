"""Where the leapfrogs of the config-5 shape go (one MI355X): per temperature rank, after a warm-up, the step sizes and the
leapfrogs per NUTS call; per wave of the kernel, the work its slowest chain dictates.  Usage: python tools/gj_census.py [pick]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmcsampler_amd.engine import PTEngine

pick = sys.argv[1] if len(sys.argv) > 1 else "chain"
d, nt, W = 20, 16, 4096
box = ("box", np.full(d, -10.0), np.full(d, 10.0))
e = PTEngine(d, nt, W, np.eye(d), logl=("curved",), logp=box, weights=(10, 0, 10), grad_weights=(10, 0), cov_update=1000, burn=10000,
             tskip=100, seed=1234, cov_mode="pooled", pick_mode=pick)
e.init_state(np.array([-0.1, -0.5] * (d // 2)))
e.run(500)
e.sync()
g0 = e.get("gj").copy()
t = time.perf_counter()
e.run(300)
e.sync()
dt = time.perf_counter() - t
g1 = e.get("gj")
calls = g1[..., 4] - g0[..., 4]
leaps = g1[..., 7] - g0[..., 7]
per = leaps / np.maximum(calls, 1)
print("pick=%s  %.3g updates/s (%.1f ms per 100 iterations)" % (pick, nt * W * 300 / dt, dt / 3 * 1e3))
print("rank   eps: median     min       max   | leapfrogs per NUTS call: median   p99     max  | share of all leapfrogs")
for r in range(nt):
    print("%3d   %10.3g %9.3g %9.3g | %28.1f %6.1f %7.1f | %.3f" % (r, np.median(g1[:, r, 0]), g1[:, r, 0].min(), g1[:, r, 0].max(),
                                                                 np.median(per[:, r]), np.percentile(per[:, r], 99), per[:, r].max(), leaps[:, r].sum() / leaps.sum()))
tot = leaps.sum()
print("leapfrogs: total %.3g; slowest chain %.3g (%.1f x the mean chain)" % (tot, leaps.max(), leaps.max() / leaps.mean()))
wave_slot = leaps.reshape(W, nt // 16, 16).max(-1).sum()              # a wave = 16 consecutive chains of a walker: pays its slowest chain
wave_rank = np.sort(leaps, axis=0).reshape(W // 16, 16, nt).max(1).sum()  # if waves held 16 walkers of one rank, sorted by load
print("wave-leapfrogs if every wave pays its slowest chain: by walker (now) %.3g = %.1f x the useful work; by rank, sorted %.3g = %.1f x"
      % (wave_slot * 16, wave_slot * 16 / tot, wave_rank * 16, wave_rank * 16 / tot))
