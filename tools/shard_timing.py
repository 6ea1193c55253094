"""Device time of one swap epoch of a sharded ladder as one of N GPUs sees it (developer tool, one GPU): block 0 of an N-block ladder,
likelihoods of the other blocks made up -- for the shapes of BASELINE configs 2, 4 and 5 as bench.py --gpus N shards them.
usage: shard_timing.py [config2 config4 config5] [--ngpus 1 2 4 8]  ->  gpurun_out/shard_timing.json (copy to profiles/rNN_shard_timing.json)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmcsampler_amd import _lib
from ptmcmcsampler_amd.engine import PTEngine

CONFIGS = {
    # bench.py's arguments for the shape                          ndim, ranks per GPU, walkers, engine keywords
    "config2": dict(bench="(defaults)", d=100, nt=64, W=4096, kw=dict(weights=(20, 0, 0))),
    "config4": dict(bench="--ndim 1000 --nwalkers 512", d=1000, nt=64, W=512, kw=dict(weights=(20, 0, 0))),
    "config5": dict(bench="--logl curved --ndim 20 --ntemps 16 --mix nuts", d=20, nt=16, W=4096,
                    kw=dict(weights=(10, 0, 10), grad_weights=(10, 0), logl=("curved",))),
}
args = sys.argv[1:]
ngpus = [1, 2, 4, 8]
if "--ngpus" in args:
    i = args.index("--ngpus")
    ngpus = [int(v) for v in args[i + 1:]]
    args = args[:i]
names = args or ["config2", "config4", "config5"]
out = {"what": "device time per swap epoch as block 0 of an N-block ladder sees it (gather, sweep of the whole ladder, pack, apply, AM row), 100 MH steps "
               "of the block's chains, the owner's pooled statistics per covariance epoch; one MI355X, tools/shard_timing.py", "configs": {}}
for name in names:
    c = CONFIGS[name]
    d, nt, W = c["d"], c["nt"], c["W"]
    kw = dict(cov_update=1000, burn=10000, tskip=100, seed=1, cov_mode="pooled", use_de_buffer=False)
    kw.update(c["kw"])
    cov0, p0 = np.eye(d) * 0.01, np.zeros(d)
    if name == "config5":
        kw.update(logp=("box", np.full(d, -10.0), np.full(d, 10.0)))
        cov0, p0 = np.eye(d), np.array([-0.1, -0.5] * (d // 2))
    results = {}
    for N in ngpus:
        ntg = nt * N
        e = PTEngine(d, nt, W, cov0, ntemps_global=ntg, temp0=0, **kw)
        e.init_state(p0)
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

        # the part's clocks need tens of milliseconds of load to come up from idle (bench.py --preheat)
        t_end = torch.cuda.Event(enable_timing=True)
        for k in range(400 if name == "config2" else (60 if name == "config4" else 6)):
            e.mh_steps(100 * k + 1, 99)
        torch.cuda.synchronize()
        epoch(100)
        ms = timed(epoch, 20)
        mh = timed(lambda it: e.mh_steps(it + 1, 99), 50 if name == "config2" else (20 if name == "config4" else 4)) * 100.0 / 99.0
        for seg in range(10):                                    # a full ring of rows and flags (no swaps: block 0 alone cannot run them)
            e.mh_steps(100 * seg + 1, 100)
        cov = timed(lambda it: _lib.check(e.lib.ptmi_update_cov(e.h, 1000)), 5 if name != "config4" else 3)
        results[str(N)] = {"swap_epoch_device_ms": ms, "mh_100_steps_ms": mh, "cov_epoch_stats_ms": cov}
        print("%s N=%d (ladder of %d): swap epoch %.3f ms on the device, 100 MH steps %.3f ms (%.1f %%), statistics %.3f ms per covariance epoch" % (
            name, N, ntg, ms, mh, 100 * ms / mh, cov), flush=True)
        del e
        torch.cuda.empty_cache()
    out["configs"][name] = {"bench_args": c["bench"], "ndim": d, "ranks_per_gpu": nt, "nwalkers": W, "by_ngpus": results}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "shard_timing.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
