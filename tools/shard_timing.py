"""Device time of one swap epoch of a sharded ladder as one of N GPUs sees it (developer tool, one GPU):
block 0 of an N x 64-rank ladder, likelihoods of the other blocks made up.  usage: shard_timing.py [N ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmcsampler_amd.engine import PTEngine

d, nt, W = 100, 64, 4096
results = {}
for N in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]:
    ntg = nt * N
    e = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=1, cov_mode="pooled",
                 use_de_buffer=False, ntemps_global=ntg, temp0=0)
    e.init_state(np.zeros(d))
    e.mh_steps(1, 99)
    dev = e.device
    lnl_loc = torch.empty((W, nt), dtype=torch.float64, device=dev)
    parts = torch.randn((N, W, nt), dtype=torch.float64, device=dev) * 5 - 50
    mp = torch.empty((W, ntg), dtype=torch.int32, device=dev)
    send = torch.zeros((N, W, d + 2), dtype=torch.float64, device=dev)
    recv = torch.zeros((N, W, d + 2), dtype=torch.float64, device=dev)

    def epoch(it):
        e.gather_lnl(lnl_loc)
        parts[0].copy_(lnl_loc)
        e.sweep_blocks(it, parts, mp)
        e.exchange_pack(mp, send)
        e.exchange_apply(recv)
        e.write_am(it)

    def timed(f, n=5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record(e.stream)
        for k in range(n):
            f(100 * (k + 1))
        ev[1].record(e.stream)
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / n

    # the part's clocks need tens of milliseconds of load to come up from idle (bench.py --preheat): five launches from a cold
    # start read 0.87 ms where a long run holds 0.72
    for k in range(400):
        e.mh_steps(100 * k + 1, 99)
    torch.cuda.synchronize()
    epoch(100)
    ms = timed(epoch, 20)
    mh = timed(lambda it: e.mh_steps(it + 1, 99), 50)
    print("N=%d (ladder of %d): swap epoch %.3f ms on the device, 100 MH steps %.3f ms -> %.1f %% of the MH time" % (N, ntg, ms, mh, 100 * ms / mh), flush=True)
    # the owner's covariance epoch (pooled statistics over the stored rows), once per covUpdate / Tskip swap epochs
    for seg in range(10):                                    # a full ring of rows and flags (no swaps: block 0 alone cannot run them)
        e.mh_steps(100 * seg + 1, 100)
    from ptmcmcsampler_amd import _lib
    cov = timed(lambda it: _lib.check(e.lib.ptmi_update_cov(e.h, 1000)), 5)
    results[N] = {"swap_epoch_device_ms": ms, "mh_100_steps_ms": mh, "cov_epoch_stats_ms": cov}
    del e
    torch.cuda.empty_cache()
import json
out = {"workload": "ndim=%d, %d ranks per GPU, %d walkers, SCAM cycle, pooled covariance (am_mode rle), Tskip=100, covUpdate=1000" % (d, nt, W),
       "what": "device time per swap epoch as block 0 of an N x 64-rank ladder sees it (gather, sweep of the whole ladder, pack, apply, AM row), "
               "100 MH steps of its 64 x 4096 chains, the owner's pooled statistics per covariance epoch; one MI355X, tools/shard_timing.py",
       "by_ngpus": {str(k): v for k, v in results.items()}}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "shard_timing.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
