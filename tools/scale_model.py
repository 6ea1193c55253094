#!/usr/bin/env python
"""Predicted weak-scaling curve of bench.py --gpus N (temperature-block partition, 64 ranks per GPU) from what CAN be measured on
one MI355X -- no multi-GPU number has ever been measured for this repository (the builder's lease is one GPU; the driver's SCALE
file is the measurement).  usage: python tools/scale_model.py [profiles/r04_shard_timing.json]

Inputs (measured, tools/shard_timing.py -> profiles/rNN_shard_timing.json): per N, the device time of one swap epoch as block 0
of an N x 64-rank ladder sees it, the time of 100 MH steps of its 64 x 4096 chains, and the owner's pooled statistics per
covariance epoch.  Assumptions (stated, not measured): xGMI is point to point, one link per neighbour,
LINK_GBS effective per direction and LINK_LAT_US per message (MI355X_MICROARCH.md: 7 links x ~153 GB/s peak per GPU; a ring
all-gather and a neighbour send/recv each use ONE link per hop).  The model per Tskip cycle of one GPU:

    t(N) = t_mh + t_swap_dev(N) + t_allgather(N) + t_edge + [owner only, per covariance epoch / 10] t_stats / 10 + t_bcast / 10

 * t_allgather: ring all-gather of lnL, (N - 1) hops of W x 64 x 8 B = 2 MB each;
 * t_edge: the grouped send/recv with the two neighbours, W x (d + 2) x 8 B = 3.3 MB each way on its own link, in parallel;
 * the owner of rank 0 runs the pooled statistics; with eig_lag = 1 nobody waits for its factorization, but every GPU meets it
   again at the next swap's all-gather, so the slowest GPU (the owner) sets the pace;
 * value(N) = N x 64 x 4096 x 100 / t(N).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK_GBS = 100.0        # effective GB/s per xGMI link and direction (peak ~153; RCCL send/recv of MB-sized messages)
LINK_LAT_US = 12.0      # per RCCL send/recv or ring step, launch included
W, NT, D = 4096, 64, 100


def predict(meas):
    rows = []
    for N in (1, 2, 4, 8):
        m = meas["by_ngpus"].get(str(N))
        if m is None:
            continue
        t_mh = m["mh_100_steps_ms"]
        t_swap = m["swap_epoch_device_ms"]
        hop = W * NT * 8 / (LINK_GBS * 1e9) * 1e3 + LINK_LAT_US * 1e-3
        t_ag = (N - 1) * hop if N > 1 else 0.0
        t_edge = (W * (D + 2) * 8 / (LINK_GBS * 1e9) * 1e3 + LINK_LAT_US * 1e-3) if N > 1 else 0.0
        t_stats = m["cov_epoch_stats_ms"] / 10.0
        t_bcast = ((D * D + D) * 8 / (LINK_GBS * 1e9) * 1e3 + LINK_LAT_US * 1e-3) * (1 if N > 1 else 0) / 10.0
        t = t_mh + t_swap + t_ag + t_edge + t_stats + t_bcast
        rows.append(dict(n_gpus=N, ms_per_step=t, updates_per_s=N * NT * W * 100 / (t * 1e-3), mh_ms=t_mh, swap_device_ms=t_swap,
                         allgather_ms=t_ag, edge_ms=t_edge, owner_stats_ms=t_stats, bcast_ms=t_bcast))
    base = rows[0]["updates_per_s"] if rows else 1.0
    for r in rows:
        r["efficiency_vs_1gpu"] = r["updates_per_s"] / (r["n_gpus"] * base)
    return rows


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else None
    if path is None:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_shard_timing.json"))
        if not cands:
            raise SystemExit("no profiles/*_shard_timing.json: run tools/shard_timing.py on a GPU first")
        path = os.path.join(ROOT, "profiles", cands[-1])
    meas = json.load(open(path))
    rows = predict(meas)
    print("# predicted from %s (LINK_GBS = %g, LINK_LAT_US = %g: assumptions, not measurements)" % (os.path.relpath(path, ROOT), LINK_GBS, LINK_LAT_US))
    print("%6s %12s %14s %8s | %8s %10s %10s %8s %10s" % ("N", "ms/step", "updates/s", "eff", "MH", "swap dev", "allgather", "edge", "stats/10"))
    for r in rows:
        print("%6d %12.3f %14.4g %8.3f | %8.3f %10.3f %10.3f %8.3f %10.3f" % (
            r["n_gpus"], r["ms_per_step"], r["updates_per_s"], r["efficiency_vs_1gpu"], r["mh_ms"], r["swap_device_ms"], r["allgather_ms"],
            r["edge_ms"], r["owner_stats_ms"]))
    json.dump({"inputs": os.path.relpath(path, ROOT), "link_gbs": LINK_GBS, "link_lat_us": LINK_LAT_US, "predicted": rows},
              open(os.path.join(ROOT, "profiles", os.path.basename(path).replace("_shard_timing", "_scale_model")), "w"), indent=1)


if __name__ == "__main__":
    main()
