#!/usr/bin/env python
"""Predicted weak-scaling curves of bench.py --gpus N (temperature-block partition) for the shapes of BASELINE configs 2, 4 and 5,
from what CAN be measured on one MI355X -- no multi-GPU number has ever been measured for this repository (the builder's lease is
one GPU; the driver's SCALE file is the measurement).  usage: python tools/scale_model.py [profiles/rNN_shard_timing.json] [config4=<bench.py ms per step on one GPU> ...]

Inputs (measured, tools/shard_timing.py -> profiles/rNN_shard_timing.json): per config and N, the device time of one swap epoch as
block 0 of an N-block ladder sees it, the time of 100 MH steps of the block's chains, and the owner's pooled statistics per
covariance epoch.  Assumptions (stated, not measured): xGMI is point to point, one link per neighbour, LINK_GBS effective per
direction and LINK_LAT_US per message (MI355X_MICROARCH.md: 7 links x ~153 GB/s peak per GPU; a ring all-gather and a neighbour
send/recv each use ONE link per hop).  The model per Tskip cycle of one GPU:

    t(N) = t_mh + t_swap_dev(N) + t_allgather(N) + t_edge + [owner, per covariance epoch / 10] (t_stats + t_bcast) / 10

 * t_allgather: the lnL gather as ShardedPTEngine's DistComm.all_gather runs it -- every GPU sends its block of W x ranks-per-GPU x 8 B
   (2 MB at config 2, 0.26 MB at config 4, 0.5 MB at config 5) to all N - 1 peers at once, one xGMI link each, in one grouped
   send/recv: ONE hop + PEER_US per further peer of the group; `ring` (round 4's model, the library's ring collective): N - 1 hops in turn;
 * t_edge: the grouped send/recv with the two neighbours, W x (d + 2) x 8 B each way on its own link, in parallel (3.3 MB at
   config 2; 4.1 MB of 8 KB rows at config 4; 0.7 MB at config 5);
 * the owner of rank 0 runs the pooled statistics (on its stream, or -- stats_async -- beside its launches: the same device time
   either way, measured); its factorization runs meanwhile (eig_lag: host LAPACK, or ptmi_eig_sytrd on its side stream) and the
   table (Ut, S: 80 KB at ndim = 100, 8 MB at ndim = 1000) is broadcast behind the L-th launch's swap; every GPU meets the owner
   again at the next swap's all-gather, so the slowest GPU (the owner) sets the pace;
 * value(N) = N x ranks-per-GPU x W x 100 / t(N); the efficiency is against the model's own N = 1 (which carries the owner's
   statistics too: weak scaling gives every GPU 64 ranks, and GPU 0 of an N = 1 run is the owner).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK_GBS = 100.0        # effective GB/s per xGMI link and direction (peak ~153; RCCL send/recv of MB-sized messages)
LINK_LAT_US = 12.0      # per RCCL send/recv or ring step, launch included
PEER_US = 2.0           # per further peer inside one grouped send/recv


def msg_ms(nbytes):
    return nbytes / (LINK_GBS * 1e9) * 1e3 + LINK_LAT_US * 1e-3


def predict(cfg, bench_ms=None):
    """bench_ms: ms per step of bench.py on ONE GPU for this shape (measured).  What it holds beyond the model's N = 1 terms -- launch
    gaps, the wait for a table that is not ready when its eig_lag launches are over (config 4: the owner's statistics + tridiagonalization
    + divide-and-conquer take longer than nine launches) -- is carried to every N as `other_ms`: the owner has it at any N."""
    W, NT, D = cfg["nwalkers"], cfg["ranks_per_gpu"], cfg["ndim"]
    rows = []
    other = 0.0
    m1 = cfg["by_ngpus"].get("1")
    if bench_ms is not None and m1 is not None:
        other = max(0.0, bench_ms - (m1["mh_100_steps_ms"] + m1["swap_epoch_device_ms"] + m1["cov_epoch_stats_ms"] / 10.0))
    # block 0 holds the same ranks and walkers whatever N is: its 100 MH steps are ONE quantity, measured once per N (configs 2 and 4
    # agree within 2 %; config 5's NUTS launches vary by +-15 % from run to run -- other streams, other trees): the mean is used
    mh_all = [m["mh_100_steps_ms"] for m in cfg["by_ngpus"].values()]
    mh_mean = sum(mh_all) / len(mh_all)
    if bench_ms is not None and m1 is not None:
        other = max(0.0, bench_ms - (mh_mean + m1["swap_epoch_device_ms"] + m1["cov_epoch_stats_ms"] / 10.0))
    for N in (1, 2, 4, 8):
        m = cfg["by_ngpus"].get(str(N))
        if m is None:
            continue
        t_mh = mh_mean
        t_swap = m["swap_epoch_device_ms"]
        t_ag_ring = (N - 1) * msg_ms(W * NT * 8) if N > 1 else 0.0
        t_ag = (msg_ms(W * NT * 8) + (N - 2) * PEER_US * 1e-3) if N > 1 else 0.0
        t_edge = msg_ms(W * (D + 2) * 8) if N > 1 else 0.0
        t_stats = m["cov_epoch_stats_ms"] / 10.0
        t_bcast = msg_ms((D * D + D) * 8) * (1 if N > 1 else 0) / 10.0
        t = t_mh + t_swap + t_ag + t_edge + t_stats + t_bcast + other
        rows.append(dict(n_gpus=N, ms_per_step=t, updates_per_s=N * NT * W * 100 / (t * 1e-3), mh_ms=t_mh, swap_device_ms=t_swap,
                         allgather_ms=t_ag, edge_ms=t_edge, owner_stats_ms=t_stats, bcast_ms=t_bcast, other_ms=other,
                         ring_allgather_ms=t_ag_ring, ms_per_step_with_ring_allgather=t - t_ag + t_ag_ring))
    base = rows[0]["updates_per_s"] if rows else 1.0
    for r in rows:
        r["efficiency_vs_1gpu"] = r["updates_per_s"] / (r["n_gpus"] * base)
        r["efficiency_with_ring_allgather"] = rows[0]["ms_per_step"] / r["ms_per_step_with_ring_allgather"]
    return rows


def main():
    argv = [a for a in sys.argv[1:] if "=" not in a]
    bench = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}     # config4=3.94: bench.py's ms per step on one GPU
    path = argv[0] if argv else None
    if path is None:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_shard_timing.json"))
        if not cands:
            raise SystemExit("no profiles/*_shard_timing.json: run tools/shard_timing.py on a GPU first")
        path = os.path.join(ROOT, "profiles", cands[-1])
    meas = json.load(open(path))
    if "configs" not in meas:                                  # round 4's file: config 2 only
        meas = {"configs": {"config2": {"bench_args": "(defaults)", "ndim": 100, "ranks_per_gpu": 64, "nwalkers": 4096, "by_ngpus": meas["by_ngpus"]}}}
    out = {"inputs": os.path.relpath(path, ROOT), "link_gbs": LINK_GBS, "link_lat_us": LINK_LAT_US, "configs": {}}
    print("# predicted from %s (LINK_GBS = %g, LINK_LAT_US = %g: assumptions, not measurements)" % (os.path.relpath(path, ROOT), LINK_GBS, LINK_LAT_US))
    for name, cfg in meas["configs"].items():
        rows = predict(cfg, bench.get(name, cfg.get("bench_1gpu_ms_per_step")))
        print("%s  (bench.py %s --gpus N: %d-d, %d ranks x %d walkers per GPU)" % (name, cfg["bench_args"], cfg["ndim"], cfg["ranks_per_gpu"], cfg["nwalkers"]))
        print("%6s %12s %14s %8s | %8s %10s %10s %8s %10s %9s %8s | %9s" % ("N", "ms/step", "updates/s", "eff", "MH", "swap dev", "allgather", "edge", "stats/10", "bcast/10", "other", "eff(ring)"))
        for r in rows:
            print("%6d %12.3f %14.4g %8.3f | %8.3f %10.3f %10.3f %8.3f %10.3f %9.4f %8.3f | %9.3f" % (
                r["n_gpus"], r["ms_per_step"], r["updates_per_s"], r["efficiency_vs_1gpu"], r["mh_ms"], r["swap_device_ms"], r["allgather_ms"],
                r["edge_ms"], r["owner_stats_ms"], r["bcast_ms"], r["other_ms"], r["efficiency_with_ring_allgather"]))
        out["configs"][name] = {"bench_args": cfg["bench_args"], "predicted": rows}
    json.dump(out, open(os.path.join(ROOT, "profiles", os.path.basename(path).replace("_shard_timing", "_scale_model")), "w"), indent=1)


if __name__ == "__main__":
    main()
