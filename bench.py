#!/usr/bin/env python
"""bench.py -- MH updates/sec of the fused Metropolis-Hastings hot path on MI355X.

Workload (BASELINE.json configs[1]): 100-d isotropic Gaussian logl,
flat prior, 64 temperatures x 4096 walkers per GPU, p0 = 0, cov0 = 0.01 I, SCAM proposal
cycle ("SCAM + accept kernel"), Tskip = 100, covUpdate = 1000, burn = 10000, seed 1234,
pooled covariance.  One "step" = one MH iteration of every chain (swap and covariance
epochs included in the wall time).  With --gpus N the ladder is sharded by temperature
block (64 ranks per GPU, 64*N in the ladder) and rows cross block edges over RCCL.

Prints ONE JSON line (rank 0).  `value` = whole-job MH updates per second with the state
resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix = FP64 vector


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--ndim", type=int, default=100)
    ap.add_argument("--ntemps", type=int, default=64, help="temperature ranks per GPU")
    ap.add_argument("--nwalkers", type=int, default=4096)
    ap.add_argument("--mix", default="scam", choices=["scam", "default", "nuts"],
                    help="scam: SCAM-only; default: SCAM/AM/DE 20/20/20; nuts: SCAM/DE/NUTS 10/10/10 (BASELINE configs[4])")
    ap.add_argument("--logl", default="iso", choices=["iso", "dense", "curved"])
    ap.add_argument("--cov-mode", default="pooled", choices=["pooled", "per_walker"])
    ap.add_argument("--swap-mode", default="sweep", choices=["sweep", "oddeven"], help="sweep: PTswap as the reference; oddeven: disjoint pairs")
    ap.add_argument("--partition", default="temps", choices=["temps", "walkers"],
                    help="N > 1: temps = one ladder of N x ntemps ranks sharded by temperature block (swap exchange over RCCL); "
                         "walkers = every GPU holds whole ladders of its own walkers (no data-path collective)")
    ap.add_argument("--sharded", action="store_true", help="use the sharded engine even with one rank (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=100000, help="iterations per usable host core of the CPU baseline (10-30 s)")
    ap.add_argument("--ess-walkers", type=int, default=32)
    return ap.parse_args()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


T0 = time.perf_counter()


def cpu_baseline(a, weights):
    """Timed in a fresh interpreter BEFORE this process touches the GPU (no fork after HIP init)."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); from oracle import numpy_port as p; "
            "v,c,w = p.time_baseline(ndim=%d, niter=%d, covUpdate=1000, burn=10000, weights=%r); "
            "print(json.dumps([v,c,w]))" % (ROOT, a.ndim, a.cpu_iters, tuple(weights)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")   # one core per chain
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-500:])
    v, cores, what = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": v, "unit": "updates/s", "cores": cores, "kind": "port",
            "sample": "reference-equivalent NumPy port (oracle/numpy_port.py), " + what}


def main():
    a = parse()
    # stdout must carry exactly one JSON line: park everything else the process (and RCCL's C-level banner)
    # prints on stderr, and keep the real stdout for the result
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    weights = {"scam": (20, 0, 0), "default": (20, 20, 20), "nuts": (10, 0, 10)}[a.mix]
    cpu = None
    # the CPU baseline is timed on rank 0 of the single-GPU run only; the NumPy port covers the Gaussian configs
    if rank == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not a.no_cpu_baseline and a.mix != "nuts" and a.logl != "curved":
        cpu = cpu_baseline(a, weights)
        log("cpu baseline %.3g updates/s on %d cores" % (cpu["value"], cpu["cores"]))
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (a.gpus, a.gpus))
    backend = os.environ.get("PTMI_DIST_BACKEND", "nccl")          # "gloo" only to rehearse N ranks on a one-GPU box
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or a.sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    d, nt, W = a.ndim, a.ntemps, a.nwalkers
    logl = ("iso",)
    if a.logl == "dense":
        A = np.random.default_rng(0).standard_normal((d, d))
        logl = ("dense", np.zeros(d), np.linalg.inv(A @ A.T / d + np.eye(d)))
    kw = dict(weights=weights, cov_update=1000, burn=10000, tskip=100, seed=1234, cov_mode=a.cov_mode, logl=logl,
              device=local, swap_mode=a.swap_mode)
    cov0, p0 = np.eye(d) * 0.01, np.zeros(d)
    if a.logl == "curved":                      # examples/curved_likelihood.ipynb: box prior [-10, 10], cov = I, start near the mode
        kw.update(logl=("curved",), logp=("box", np.full(d, -10.0), np.full(d, 10.0)))
        cov0, p0 = np.eye(d), np.array([-0.1, -0.5] * (d // 2) + [0.0] * (d % 2))
    if a.mix == "nuts":
        kw.update(grad_weights=(10, 0))
    if (world == 1 and not a.sharded) or a.partition == "walkers":
        from ptmcmcsampler_amd.engine import PTEngine
        eng = PTEngine(d, nt, W, cov0, walker0=rank * W, **kw)       # distinct RNG streams per GPU
    else:
        from ptmcmcsampler_amd.sharded import ShardedPTEngine
        eng = ShardedPTEngine(d, nt * world, W, cov0, group=dist.group.WORLD, **kw)
    eng.init_state(p0)
    log("engine ready")

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    eng.run(a.warmup)
    fence()
    log("warmup done")
    # timed region: exactly --steps iterations; each fused-MH launch is bracketed by HIP events on the
    # engine's stream (= torch's current stream, the one the kernels are launched on)
    events = []
    orig = eng.mh_steps

    def timed_mh(iter0, nsteps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        orig(iter0, nsteps)
        e1.record(eng.stream)
        events.append((e0, e1, nsteps))

    eng.mh_steps = timed_mh
    ess_keep = []
    nw_ess = min(a.ess_walkers, W) if eng.owns_cold else 0
    orig_cov = eng.update_cov

    def cov_and_keep(it_done):
        if nw_ess:
            ess_keep.append(eng.t["AM"][:nw_ess].clone())       # device-side copy of the cold samples of a few walkers
        orig_cov(it_done)

    eng.update_cov = cov_and_keep
    t0 = time.perf_counter()
    eng.run(a.steps)
    fence()
    wall = time.perf_counter() - t0
    eng.mh_steps, eng.update_cov = orig, orig_cov
    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())
    log("timed region %.3f s" % wall)

    nchains_total = nt * world * W
    value = nchains_total * a.steps / wall
    kern_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in events)
    kern_steps = sum(n for _, _, n in events)
    bytes_per_update = 16 * d + 32
    avg_launch_ms = kern_ms / max(1, len(events))
    avg_steps = kern_steps / max(1, len(events))
    achieved = bytes_per_update * nt * W * avg_steps / (avg_launch_ms * 1e-3) / 1e9
    out = {
        "metric": "MH updates/sec (whole node), 100-d Gaussian, 64 temps x 4096 walkers per GPU",
        "value": value, "unit": "updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: %d-d %s logl, %d temps x %d walkers per GPU, %s cycle, "
                               "Tskip=100 (%s), covUpdate=1000, cov_mode=%s" % (
                                   {"iso": 3 if d >= 1000 else 1, "dense": 2, "curved": 4}[a.logl], d,
                                   {"iso": "isotropic Gaussian", "dense": "dense Gaussian", "curved": "curved-likelihood"}[a.logl],
                                   nt, W, a.mix, a.swap_mode, a.cov_mode),
                   "ndim": d, "ntemps_per_gpu": nt, "nwalkers": W, "parallelism": ("temperature blocks x%d" if a.partition == "temps" else "walker blocks x%d") % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "mh_steps_gj_kernel" if a.mix == "nuts" else "mh_steps_kernel", "avg_launch_ms": avg_launch_ms, "steps_per_launch": avg_steps,
                     "algorithmic_bytes_per_update": bytes_per_update, "kernel_time_share_of_wall": kern_ms * 1e-3 / wall},
    }
    # HBM traffic of the dominant kernel comes from separate rocprofv3 PMC passes (tools/gpu_profile.sh); the committed
    # summary applies only to the exact workload it was measured on
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        key = "ndim=%d ntemps=%d nwalkers=%d mix=%s logl=%s steps_per_launch=%d" % (d, nt, W, a.mix, a.logl, int(round(avg_steps)))
        if tr.get("workload") == key:
            out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
            out["roofline"]["traffic_note"] = "HBM bytes per launch from rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, " + ", ".join(tr["source"])
            out["roofline"]["algorithmic_bytes_per_launch"] = bytes_per_update * nt * W * avg_steps
    except (OSError, ValueError):
        pass
    if a.logl == "dense":
        # config 3: the dense contraction bounds the kernel: 2d^2+3d flop per likelihood, +2d^2 per AM proposal
        flops = 2 * d * d + 3 * d + (2 * d * d * weights[1] / float(sum(weights)) if weights[1] else 0.0)
        tf = flops * nt * W * avg_steps / (avg_launch_ms * 1e-3) / 1e12
        out["roofline"].update({"bound": "mfma", "achieved": tf, "peak": F64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tf / F64_MATRIX_PEAK_TFLOPS, "algorithmic_flops_per_update": flops})
    if rank == 0:
        acc = eng.get("nacc").astype(np.float64)
        out["acceptance_rank0_mean"] = float(acc[:, 0].mean() / (a.steps + a.warmup))
        out["swap_accept_rate_pair0"] = float(eng.get("nswap")[:, 0].mean() / max(1, eng.swap_proposed))
        if ess_keep:
            from ptmcmcsampler_amd.ess import ess
            blocks = [b.cpu().numpy() for b in ess_keep]
            cu = blocks[0].shape[1]
            # AM rows are in ring order (row 0 = newest); restore time order before concatenating
            chain = np.concatenate([np.concatenate([b[:, 1:], b[:, :1]], axis=1) for b in blocks], axis=1)
            per_walker = [ess(chain[w]) for w in range(chain.shape[0])]
            covered = cu * len(blocks)
            out["ess_per_sec"] = float(np.mean(per_walker) / covered * a.steps * W / wall)
            out["ess_note"] = "Sokal-window ESS (min over dims) of the T=1 chain, mean over %d walkers x %d samples, scaled to %d walkers" % (
                len(per_walker), covered, W)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            # the C oracle on one core, for scale (a compiled scalar port; not what a reference user gets)
            from oracle import oracle as orc
            o = orc.OracleEngine(d, 8, 8, np.eye(d) * 0.01, weights=weights, cov_update=1000, burn=10000, tskip=100,
                                 seed=1234, cov_mode=a.cov_mode)
            o.init_state(np.zeros(d))
            t1 = time.perf_counter()
            o.run(1000)
            out["cpu_c_oracle"] = {"value": 64 * 1000 / (time.perf_counter() - t1), "unit": "updates/s", "cores": 1,
                                   "sample": "oracle/ptmcmc_oracle.c, 8 temps x 8 walkers x 1000 iterations"}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
