#!/usr/bin/env python
"""bench.py -- MH updates/sec (+ ESS/sec) of the fused Metropolis-Hastings hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): 100-d isotropic Gaussian logl, flat prior, 64 temperatures x 4096
walkers per GPU, p0 = 0, cov0 = 0.01 I, SCAM proposal cycle ("SCAM + accept kernel"), Tskip = 100, covUpdate = 1000,
burn = 10000, seed 1234, pooled covariance.

One "step" = one Tskip cycle of the hot path over the whole batch: 100 Metropolis-Hastings iterations of every chain, the
PT swap that closes the cycle, and whatever covariance / DE epochs fall inside (one covariance epoch every 10 steps).
`--steps K --warmup W` therefore time K*100 iterations after W*100 untimed ones; swap and covariance epochs are inside the
wall time whatever K is (K >= 10 contains at least one covariance epoch).  `value` = chains x iterations / wall, whole job,
with the state resident in HBM before the timed region.

`python bench.py --gpus N` launches its own N ranks (torch.distributed.run, one per GPU, RCCL); under an external
`python -m torch.distributed.run ... bench.py --gpus N` it uses the ranks it is given.  The ladder is then sharded by
temperature block (64 ranks per GPU, 64*N in the ladder) and rows cross block edges over RCCL.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_PEAK_TFLOPS = 78.6         # MI355X FP64 vector = FP64 matrix peak
F64_MFMA_SUSTAINED = 47.9      # what back-to-back v_mfma_f64_16x16x4 instructions reach on this part (tools/mfma_peak.hip): 0.61 of nominal
TSKIP = 100                    # iterations per step


def make_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed Tskip cycles (100 MH iterations + swap each)")
    ap.add_argument("--warmup", type=int, default=100, help="untimed Tskip cycles before them")
    ap.add_argument("--ndim", type=int, default=100)
    ap.add_argument("--ntemps", type=int, default=64, help="temperature ranks per GPU")
    ap.add_argument("--nwalkers", type=int, default=4096)
    ap.add_argument("--mix", default="scam", choices=["scam", "default", "nuts"],
                    help="scam: SCAM-only; default: SCAM/AM/DE 20/20/20; nuts: SCAM/DE/NUTS 10/10/10 (BASELINE configs[4])")
    ap.add_argument("--weights", default=None, help="SCAM,AM,DE weights overriding --mix (kernel experiments)")
    ap.add_argument("--pick", default="chain", choices=["chain", "walker"],
                    help="chain: every chain picks its proposal from its own stream (a replica of the reference); "
                         "walker: one pick per walker and iteration (wave-uniform proposal type)")
    ap.add_argument("--logl", default="iso", choices=["iso", "dense", "curved"])
    ap.add_argument("--prior", default="flat", choices=["flat", "box"],
                    help="flat: the headline workload; box: uniform on [-10, 10]^d, the usual lnpriorfn of a reference run")
    ap.add_argument("--cov-mode", default="pooled", choices=["pooled", "pooled_device", "pooled_hipsolver", "pooled_sytrd", "per_walker", "per_walker_device", "per_walker_jacobi"],
                    help="pooled: one covariance from all walkers; per_walker: every walker adapts its own (a replica of a reference run); "
                         "_device: the covariance epochs are factorized on the device (tridiagonal QL kernel) instead of host LAPACK, "
                         "_jacobi: by the device Jacobi kernel; "
                         "_hipsolver: by the ROCm library's eigensolver on the stream (large ndim)")
    ap.add_argument("--eig-lag", type=int, default=-1,
                    help="pooled covariance: the eigenvectors of a covariance epoch take effect this many launches late, the factorization "
                         "running meanwhile (PTEngine eig_lag: the host's LAPACK beside the GPU, or the ROCm library on a side stream); "
                         "0 = at once, the GPU idle / the stream blocked meanwhile (the reference's order); default: 1 with the host's LAPACK, "
                         "10 (a covariance period) with the device factorizations (ndim >= 512)")
    ap.add_argument("--am-mode", default="auto", choices=["auto", "rows", "rle"],
                    help="how the rank-0 chain's samples are kept between covariance epochs (PTEngine am_mode): rle = a step stores its "
                         "row only when it was accepted, the pooled statistics weight every stored row by its run length; rows = every "
                         "step stores its row")
    ap.add_argument("--stats-async", default="off", choices=["on", "off"],
                    help="on: with eig_lag >= 1 the pooled statistics of a finished covariance period run on a side stream beside the next "
                         "period's launches (two AM rings, PTEngine stats_async); off (default): on the engine's stream.  Measured (round 5): "
                         "no gain on one GPU -- config 2's persistent step blocks leave no room for a statistics block on their CUs and the "
                         "host's factorization then starts late (2.4e10 against 3.0e10 in the driver's window), config 4's launches slow "
                         "down by what the statistics take (8.3e8 either way)")
    ap.add_argument("--swap-mode", default="sweep", choices=["sweep", "oddeven"], help="sweep: PTswap as the reference; oddeven: disjoint pairs")
    ap.add_argument("--partition", default="temps", choices=["temps", "walkers"],
                    help="N > 1: temps = one ladder of N x ntemps ranks sharded by temperature block (swap exchange over RCCL); "
                         "walkers = every GPU holds whole ladders of its own walkers (no data-path collective)")
    ap.add_argument("--sharded", action="store_true", help="use the sharded engine even with one rank (testing)")
    ap.add_argument("--callback", action="store_true",
                    help="evaluate the likelihood in a batched torch callback between ptmi_propose and ptmi_accept (one launch pair per "
                         "iteration) instead of inside the fused kernel")
    ap.add_argument("--callback-kind", default="norm", choices=["norm", "naive", "hip"],
                    help="the callback: norm = torch, -0.5 * vector_norm(X)^2 (one pass over the proposals); naive = torch, "
                         "-0.5 * (X * X).sum(-1) (writes and re-reads a temporary of the proposals' size); hip = a device kernel behind the C "
                         "ABI (ptmi_rows_logl: the fused kernels' likelihood bits)")
    ap.add_argument("--callback-graph", action="store_true",
                    help="--callback: every segment (one proposal launch, then per iteration the callback and ptmi_accept_propose) captured "
                         "once in a hipGraph and replayed -- for small, launch-bound batches (PTEngine.callback_segment_graph)")
    ap.add_argument("--callback-launches", default="one", choices=["one", "two"],
                    help="one: ptmi_accept_propose (the accept test and the next proposal in one launch); two: ptmi_propose + ptmi_accept")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also-child", action="store_true", help=argparse.SUPPRESS)      # the process that runs the "also" legs (spawned by the headline's)
    ap.add_argument("--also", default="auto", choices=["auto", "on", "off"],
                    help="after the headline's timed region (untouched by them), short legs of the other BASELINE configs on the same GPU, "
                         "reported under \"also\": dense SCAM (configs[2]), default mix, dense default mix, a replica of reference runs "
                         "(per-walker covariance, device QL), config 4's share of one GPU, config 5's share.  auto = when the headline is "
                         "the default workload on one GPU")
    ap.add_argument("--preheat", type=float, default=0.2,
                    help="seconds of unrelated f64 matrix products before the warmup steps, so that the timed steps run at the clocks "
                         "of a long run (0 = from cold clocks)")
    ap.add_argument("--cpu-iters", type=int, default=100000, help="iterations per usable host core of the CPU baseline (10-30 s)")
    ap.add_argument("--ess-walkers", type=int, default=16, help="walkers whose T=1 chains the autocorrelation time is estimated from")
    ap.add_argument("--ess-burn", type=int, default=40000, help="the ESS window starts at this iteration at the earliest (adaptation settled)")
    ap.add_argument("--ess-window", type=int, default=80000, help="iterations of the ESS window, run AFTER the timed region (0 = no ESS)")
    ap.add_argument("--ess-max-seconds", type=float, default=60.0, help="the ESS leg is shortened to fit this (flagged when < 50 tau)")
    return ap


def parse():
    return make_parser().parse_args()


def parse_defaults():
    return make_parser().parse_args([])


T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, and hand their exit
    code on.  Rank 0 of the children prints the JSON line on the inherited stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="1")
    return subprocess.call(cmd, env=env)


def cpu_baseline(a, weights):
    """Timed in a fresh interpreter BEFORE this process touches the GPU (no fork after HIP init)."""
    code = ("import json,sys; sys.path.insert(0, %r); from oracle import numpy_port as p; "
            "v,c,w = p.time_baseline(ndim=%d, niter=%d, covUpdate=1000, burn=10000, weights=%r); "
            "print(json.dumps([v,c,w]))" % (ROOT, a.ndim, a.cpu_iters, tuple(weights)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")   # one core per chain
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-500:])
    v, cores, what = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": v, "unit": "updates/s", "cores": cores, "kind": "port",
            "sample": "reference-equivalent NumPy port (oracle/numpy_port.py; per-core rate within 10 % of the reference itself, "
                      "oracle/baseline_calibration.json), one chain per core at the ladder's first temperatures, no swaps: " + what}


class ColdSamples(object):
    """T = 1 samples of a few walkers over the timed region, copied on the device from the AM ring (which holds the last
    covUpdate cold samples of every walker, row = iteration % covUpdate) whenever the ring is about to wrap."""

    def __init__(self, eng, nw):
        self.eng, self.nw, self.snaps = eng, nw, []

    def snap(self, it_done):
        if self.nw:
            self.eng.am_expand(0, self.nw, it_hi=it_done)             # the repeats of rejected steps copied forward for the walkers kept (a no-op with stored rows)
            self.snaps.append((it_done, self.eng.t["AM"][:self.nw].clone()))

    def series(self, first, last):
        """[nw][last - first + 1][d], iterations first..last in time order."""
        import numpy as np
        cu = self.eng.cov_update
        out = None
        have = np.zeros(last - first + 1, dtype=bool)
        for it_done, ring in self.snaps:
            ring = self.eng.am_params(ring.cpu().numpy())             # parameter order whatever the buffer's row format
            if out is None:
                out = np.zeros((ring.shape[0], last - first + 1, ring.shape[2]))
            its = np.arange(max(first, it_done - cu + 1), min(last, it_done) + 1)
            if len(its):
                out[:, its - first] = ring[:, its % cu]
                have[its - first] = True
        return out[:, have] if out is not None else None


def cycle_weights(a):
    """(SCAM, AM, DE) weights of the run's proposal cycle; --weights overrides --mix (and names the mix after itself)."""
    if a.weights:
        w = tuple(int(v) for v in a.weights.split(","))
        a.mix = "w" + a.weights.replace(",", "-")
        return w
    if a.mix.startswith("w") and a.mix[1:].replace("-", "").isdigit():
        return tuple(int(v) for v in a.mix[1:].split("-"))
    return {"scam": (20, 0, 0), "default": (20, 20, 20), "nuts": (10, 0, 10)}[a.mix]


def measure(a, rank, world, local, dist, backend):
    """One workload on the ranks that are up: build the engine, W untimed steps, exactly K timed steps (barrier + synchronize on
    both sides, HIP events around every fused-MH launch), the ESS leg when a.ess_window > 0.  Returns the JSON line's dict."""
    import numpy as np
    import torch
    weights = cycle_weights(a)
    d, nt, W = a.ndim, a.ntemps, a.nwalkers
    if a.cov_mode == "pooled" and d >= 512:
        # one ndim x ndim factorization per epoch: from here on the device beats the host's LAPACK -- ptmi_eig_sytrd (tridiagonalization
        # in one kernel + the library's divide-and-conquer) up to 1024, the ROCm library's eigensolver beyond
        a.cov_mode = "pooled_sytrd" if d <= 1024 else "pooled_hipsolver"
    logl = ("iso",)
    if a.logl == "dense":
        A = np.random.default_rng(0).standard_normal((d, d))
        logl = ("dense", np.zeros(d), np.linalg.inv(A @ A.T / d + np.eye(d)))
    kw = dict(weights=weights, cov_update=1000, burn=10000, tskip=TSKIP, seed=1234, logl=logl, device=local, swap_mode=a.swap_mode,
              pick_mode=a.pick, cov_mode="per_walker" if a.cov_mode.startswith("per_walker") else "pooled", am_mode=a.am_mode,
              eig_mode="ql" if a.cov_mode.endswith("_device") else ("jacobi" if a.cov_mode.endswith("_jacobi") else (
                  "hipsolver" if a.cov_mode.endswith("_hipsolver") else ("sytrd" if a.cov_mode.endswith("_sytrd") else "lapack"))))
    eig_lag = 0
    if kw["cov_mode"] == "pooled" and kw["eig_mode"] in ("lapack", "hipsolver", "sytrd"):
        # device factorizations: a whole covariance period of launches (10); the pending table is then finished behind the next
        # epoch's statistics, which run beside the rest of it (config 4: statistics 12 ms + tridiagonalization 12 + divide and conquer
        # 13 beside nine launches of 2.2 ms left the stream waiting 5 ms per period at eig_lag 9)
        eig_lag = a.eig_lag if a.eig_lag >= 0 else (1 if kw["eig_mode"] == "lapack" else 10)
    elif kw["eig_mode"] == "ql" and a.gpus == 1:
        # the device QL of every walker's covariance on a side stream beside the launches that follow (--eig-lag L): measured, no gain --
        # 1.07e10 at every lag, the step launches slow down by what the factorization takes (0.87 -> 1.77 ms): its kernels are LDS
        # traffic, not idle latency.  The default stays 0: in the replica mode a walker applies its table at once, as the reference does.
        eig_lag = a.eig_lag if a.eig_lag >= 0 else 0
    kw.update(eig_lag=eig_lag)
    # the statistics of a finished covariance period on a side stream beside the launches that follow (PTEngine stats_async: two AM
    # rings; needs the late table, and burn a multiple of covUpdate when a DE history is kept -- 10000 / 1000 here)
    stats_async = eig_lag >= 1 and a.stats_async == "on" and not a.callback
    kw.update(stats_async=stats_async)
    cov0, p0 = np.eye(d) * 0.01, np.zeros(d)
    if a.prior == "box":
        kw.update(logp=("box", np.full(d, -10.0), np.full(d, 10.0)))
    if a.logl == "curved":                      # examples/curved_likelihood.ipynb: box prior [-10, 10], cov = I, start near the mode
        kw.update(logl=("curved",), logp=("box", np.full(d, -10.0), np.full(d, 10.0)))
        cov0, p0 = np.eye(d), np.array([-0.1, -0.5] * (d // 2) + [0.0] * (d % 2))
    if a.mix == "nuts":
        kw.update(grad_weights=(10, 0))
    if (world == 1 and not a.sharded) or a.partition == "walkers":
        from ptmcmcsampler_amd.engine import PTEngine
        eng = PTEngine(d, nt, W, cov0, walker0=rank * W, split=a.callback, **kw)       # distinct RNG streams per GPU
    else:
        from ptmcmcsampler_amd.sharded import ShardedPTEngine
        eng = ShardedPTEngine(d, nt * world, W, cov0, group=dist.group.WORLD, **kw)
    if a.callback:
        if a.logl != "iso" or world != 1:
            raise SystemExit("--callback times the iso-Gaussian through a torch callback on one GPU")
        if a.callback_kind == "naive":
            cb_l = lambda X: -0.5 * (X * X).sum(-1)                  # noqa: E731
        elif a.callback_kind == "hip":
            cb_l = eng.builtin_logl()
        else:
            cb_l = lambda X: torch.linalg.vector_norm(X, dim=-1).square_().mul_(-0.5)     # noqa: E731 -- one pass over the proposals
        cb_p = None                                                  # the flat prior: no launch
        eng.init_state_callback(p0, cb_l, cb_p)
        eng.run = lambda n: eng.run_callback(n, cb_l, cb_p, fused=a.callback_launches == "one", graph=a.callback_graph)
    else:
        eng.init_state(p0)
    log("engine ready")

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    it_warm, it_timed = a.warmup * TSKIP, a.steps * TSKIP
    if a.preheat > 0:
        # The part's clocks need tens of milliseconds of load to come up from idle; a bench step is ONE millisecond, so --warmup 5
        # ends while they still ramp (measured: the first timed launches at 0.89 ms against 0.79 with the clocks up, 7 % of a
        # 20-step wall).  Unrelated f64 work brings them up BEFORE the W warmup steps; the engine's state, the W warmup steps and
        # the K timed steps are untouched.  --preheat 0 measures from cold clocks.
        hm = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
        t_h = time.perf_counter()
        while time.perf_counter() - t_h < a.preheat:
            for _ in range(8):
                hm @ hm
            torch.cuda.synchronize()
        del hm
    eng.run(it_warm)
    fence()
    log("warmup done (%d iterations)" % it_warm)
    # timed region: exactly --steps Tskip cycles; each fused-MH launch is bracketed by HIP events on the engine's
    # stream (= torch's current stream, the one the kernels are launched on)
    events = []
    # (callback path: a segment = one proposal launch, then per iteration the torch callback and ptmi_accept_propose; or split_step's
    # launch pair with --callback-launches two)
    hot = (("callback_segment_graph" if a.callback_graph else "callback_segment") if a.callback_launches == "one" else "split_step") if a.callback else "mh_steps"
    orig = getattr(eng, hot)

    def timed_mh(iter0, *rest):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        ret = orig(iter0, *rest)
        e1.record(eng.stream)
        if not (hot == "callback_segment_graph" and ret is False):
            events.append((e0, e1, (rest[0] - iter0 + 1 if hot.startswith("callback_segment") else 1) if a.callback else rest[0]))
        return ret

    setattr(eng, hot, timed_mh)
    orig_cov = eng.update_cov
    n_cov = [0]

    def cov_and_count(it_done):
        n_cov[0] += 1
        orig_cov(it_done)

    eng.update_cov = cov_and_count
    def gj_counts():
        """(leapfrogs, NUTS calls, HMC calls) so far, summed over the ranks (the jump objects' own counters)."""
        from ptmcmcsampler_amd import _lib as L
        gj = eng.t.get("gj") if hasattr(eng, "t") else None
        if gj is None:
            return None
        return tuple(float(gj[..., k].sum().item()) for k in (L.GJ_NLEAP, L.GJ_NITER, L.GJ_HITER))

    gj0 = gj_counts() if a.mix == "nuts" else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(it_timed)
    fence()
    wall = time.perf_counter() - t0
    gj1 = gj_counts() if a.mix == "nuts" else None
    setattr(eng, hot, orig)
    eng.update_cov = orig_cov
    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())
    log("timed region %.3f s (%d iterations, %d covariance epochs)" % (wall, it_timed, n_cov[0]))
    # share of the rank-0 rows the timed launches stored (am_mode rle: accepted steps and KEY rows only), read off the AM flags NOW:
    # the ring holds the last covUpdate iterations of the timed region.  (Read after the ESS leg -- at the stationary acceptance --
    # it made the byte model of the driver's window, where nearly every proposal is accepted, 27 % too small: round 4's
    # traffic_over_model 1.27.)
    stored_timed = None
    if getattr(eng, "am_rle", False) and getattr(eng, "owns_cold", False) and hasattr(eng, "t") and eng.t.get("AMflag") is not None:
        stored_timed = float(((eng.t["AMflag"] & 3) != 0).double().mean().item())

    # ---- ESS leg, OUTSIDE the timed wall.  ESS/sec = (walkers x iterations/s of the timed region) / tau_int, with the
    # integrated autocorrelation time of the T = 1 chains estimated where it can be: on a stationary window of >= 50 tau
    # after the adaptation has settled (a window inside a 2000-iteration timed region that starts 500 iterations after
    # p0 = 0 measures the transient, not the sampler).  Every rank runs the leg (the swaps are collective); rank 0 reports.
    ess_out = {"ess_per_sec": None}
    if a.ess_window > 0:
        rate = it_timed / wall                                        # iterations/s of every chain
        it_now = it_warm + it_timed
        burn_more = max(0, a.ess_burn - it_now)
        window = a.ess_window
        budget = a.ess_max_seconds * rate
        if burn_more + window > budget:                               # slow workloads: keep the leg bounded, say so
            window = int(max(2000, min(window, budget * 2 / 3)))
            burn_more = int(max(0, min(burn_more, budget - window)))
        window = (window // 1000) * 1000 or 1000
        eng.run(burn_more)
        first = it_now + burn_more + 1
        cold = ColdSamples(eng, min(a.ess_walkers, W) if eng.owns_cold else 0)

        def cov_and_keep(it_done):
            cold.snap(it_done)                  # device-side copy of a few walkers' cold samples, before the ring wraps
            orig_cov(it_done)

        eng.update_cov = cov_and_keep
        eng.run(window)
        fence()
        eng.update_cov = orig_cov
        cold.snap(first + window - 1)
        log("ESS leg done (%d + %d iterations)" % (burn_more, window))
        if rank == 0 and eng.owns_cold:
            from ptmcmcsampler_amd.ess import MIN_TAUS, integrated_time
            chain = cold.series(first, first + window - 1)            # [nw][N][d]
            taus, ok, drift = [], True, []
            for wv in chain:
                r = integrated_time(wv, full=True)
                taus.append(float(np.max(r["tau"])))
                ok = ok and bool(np.all(r["reliable"]))
                h = wv.shape[0] // 2
                drift.append(float((wv[h:] ** 2).sum(1).mean() / (wv[:h] ** 2).sum(1).mean()))
            tau_int = 1.0 / float(np.mean(1.0 / np.asarray(taus)))    # mean over walkers of ESS / N = 1 / tau
            ess_out = {
                "ess_per_sec": W * (world if a.partition == "walkers" else 1) * rate / tau_int,
                "tau_int": tau_int, "tau_int_max_walker": float(np.max(taus)), "ess_window_iters": int(chain.shape[1]),
                "ess_window_first_iter": int(first), "ess_window_ok": bool(ok and chain.shape[1] >= MIN_TAUS * np.max(taus)),
                "ess_window_taus": float(chain.shape[1] / np.max(taus)),
                "ess_stationarity_r2_second_over_first_half": float(np.mean(drift)),
                "ess_note": ("ESS/sec = walkers x iterations/s of the timed region / tau_int; tau_int = Sokal-window integrated autocorrelation "
                             "time (max over the %d parameters) of the T=1 chains of %d walkers over %d iterations from iteration %d on, run "
                             "after the timed region; ess_window_ok = the window holds >= %d tau for every walker and parameter"
                             % (d, chain.shape[0], chain.shape[1], first, MIN_TAUS)),
            }
    nchains_total = nt * world * W
    value = nchains_total * it_timed / wall
    kern_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in events)
    if os.environ.get("PTMI_BENCH_LAUNCHES"):                        # developer switch: every timed launch
        log("launch ms: " + " ".join("%.3f" % e0.elapsed_time(e1) for e0, e1, _ in events))
    kern_steps = sum(n for _, _, n in events)
    avg_launch_ms = kern_ms / max(1, len(events))
    avg_steps = kern_steps / max(1, len(events))
    upd_per_launch = nt * W * avg_steps
    bytes_per_update = 16 * d + 32
    flops_per_update = 4 * d
    hbm_view = bytes_per_update * upd_per_launch / (avg_launch_ms * 1e-3) / 1e9
    tf = flops_per_update * upd_per_launch / (avg_launch_ms * 1e-3) / 1e12
    kernel = "mh_steps_gj_kernel" if a.mix == "nuts" else (
        ("split_rows_kernel<acc,prop> + torch callback" if a.callback_launches == "one" else "split_rows_kernel<prop> + torch callback + split_rows_kernel<acc>")
        if a.callback else "mh_steps_kernel")
    out = {
        "metric": "MH updates/sec (whole node) + ESS/sec, 100-d Gaussian, 64 temps x 4096 walkers per GPU",
        "value": value, "unit": "updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "preheat_s": a.preheat,
        "config": {"workload": "BASELINE configs[%d]: %d-d %s logl%s, %d temps x %d walkers per GPU, %s cycle (pick per %s), "
                               "Tskip=100 (%s), covUpdate=1000, cov_mode=%s, am_mode=%s, eig_lag=%d%s; one step = 100 MH iterations of every chain + "
                               "the PT swap (+ a covariance epoch every 10 steps)" % (
                                   {"iso": 3 if d >= 1000 else 1, "dense": 2, "curved": 4}[a.logl], d,
                                   {"iso": "isotropic Gaussian", "dense": "dense Gaussian", "curved": "curved-likelihood"}[a.logl],
                                   " + box prior" if a.prior == "box" else "", nt, W, a.mix, a.pick, a.swap_mode, a.cov_mode,
                                   "rle" if getattr(eng, "am_rle", False) else "rows", eig_lag,
                                   ", statistics on a side stream" if getattr(eng, "stats_async", False) or getattr(getattr(eng, "local", None), "stats_async", False) else ""),
                   "ndim": d, "ntemps_per_gpu": nt, "nwalkers": W, "iterations_per_step": TSKIP,
                   "parallelism": ("temperature blocks x%d" if a.partition == "temps" else "walker blocks x%d") % world},
        "iterations_timed": it_timed, "swap_epochs_timed": it_timed // TSKIP if nt * world > 1 else 0, "cov_epochs_timed": n_cov[0],
        "rccl_ranks": world if (dist is not None and backend == "nccl") else (0 if dist is not None else 1),
        # The fused K-step kernel keeps a chain's state in registers, so it is bounded by f64 vector issue, not by HBM:
        # achieved = SURVEY 8(d)'s 4d flop per update x updates per launch / the launch's HIP-event time.
        "roofline": {"bound": "f64_valu", "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_PEAK_TFLOPS,
                     "traffic": None, "kernel": kernel, "avg_launch_ms": avg_launch_ms, "steps_per_launch": avg_steps,
                     "algorithmic_flops_per_update": flops_per_update, "algorithmic_bytes_per_update": bytes_per_update,
                     "algorithmic_hbm_gbs": hbm_view, "algorithmic_hbm_ratio": hbm_view / HBM_PEAK_GBS,
                     "kernel_time_share_of_wall": kern_ms * 1e-3 / wall,
                     # the timed launches in order: first and last (a ramp between them = clocks or adaptation still moving)
                     "launch_ms_first": events[0][0].elapsed_time(events[0][1]) if events else None,
                     "launch_ms_last": events[-1][0].elapsed_time(events[-1][1]) if events else None},
    }
    if events:
        ms_all = [e0.elapsed_time(e1) for e0, e1, _ in events]
        out["roofline"]["launch_ms_min"], out["roofline"]["launch_ms_max"] = min(ms_all), max(ms_all)
    if a.mix == "nuts" and gj0 is not None and gj1 is not None:
        # Config 5: the work of a gradient-jump cycle is its leapfrogs (nutsjump.py:149-169), counted by the jump objects themselves
        # (gj[.][GJ_NLEAP]).  Per leapfrog and parameter: 6 flop of the integrator, 4 of the diagonal whitening and
        # tempering, ~65 of the curved likelihood's value and gradient (two exp, one log, one division per PAIR of parameters, at their
        # polynomial lengths), 4 of the kinetic energy and the box test: ~79 d flop.  The other picks (SCAM / DE) are priced at 4 d.
        leaps, nuts, hmcs = (b1 - b0 for b0, b1 in zip(gj0, gj1))
        f_leap = 79.0 * d
        total_flop = f_leap * leaps + flops_per_update * (nchains_total * it_timed - nuts - hmcs)
        tfg = total_flop / (kern_ms * 1e-3) / 1e12
        out["roofline"].update({"achieved": tfg, "frac": tfg / F64_PEAK_TFLOPS, "leapfrogs_timed": leaps, "nuts_calls_timed": nuts,
                                "leapfrogs_per_nuts_call": leaps / max(1.0, nuts), "flops_per_leapfrog": f_leap,
                                "algorithmic_flops_per_update": total_flop / (nchains_total * it_timed),
                                "flops_note": "counted work: 79 d flop per leapfrog x the leapfrogs the jump objects counted in the timed region "
                                              "+ 4 d per SCAM / DE update, over the launches' HIP-event time; the launch lasts as long as its slowest chain"})
        de_in = bool(getattr(eng, "de_on", False))
        out["config"]["workload"] += "; cycle in the timed region: SCAM + NUTS%s (DE joins after burn = %d iterations)" % (" + DE" if de_in else "", 10000)
    if a.callback:
        out["config"]["workload"] += ("; the likelihood OUTSIDE the library: a batched torch callback (%s) on the device tensor of proposals, %s per iteration"
                                      % ({"norm": "torch: -0.5 * vector_norm(Q)^2, one pass", "naive": "torch: -0.5 * (Q * Q).sum(-1)", "hip": "a HIP kernel behind the C ABI: ptmi_rows_logl"}[a.callback_kind],
                                         ("ptmi_accept_propose (one launch)" + (", every segment one hipGraph launch" if a.callback_graph else "")) if a.callback_launches == "one" else "ptmi_propose + ptmi_accept (two launches)"))
        # The split path IS bound by HBM: SURVEY 8(d)'s 16 d + 32 bytes per update are real traffic here (state in, proposal out;
        # the callback's own read of the proposals and the accepted rows written back come on top: the split design's byte model)
        acc_t = None
        try:
            acc_t = float(eng.get("nacc").astype(np.float64).mean() / max(1, eng.iter))
        except Exception:               # noqa: BLE001
            pass
        # state or proposal in, proposal out, the callback's read of it; a row goes to X only when its buffer is about to be overwritten:
        # accepted one iteration ago and refused now (two proposal buffers, csrc/ptmi_split.hip) -- or every accepted row with one launch pair
        a_ = acc_t if acc_t is not None else 1.0
        model = (3.0 + (a_ * (1.0 - a_) if a.callback_launches == "one" else a_)) * 8 * d + 64 + 32          # + qaux in / out, lnL / lp / callback values
        out["roofline"].update({"bound": "hbm", "achieved": hbm_view, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_view / HBM_PEAK_GBS,
                                "f64_valu_frac": tf / F64_PEAK_TFLOPS,
                                "split_design_bytes_per_update": model, "acceptance_whole_run": acc_t,
                                "split_design_gbs": model * upd_per_launch / (avg_launch_ms * 1e-3) / 1e9,
                                "note": "achieved = (16 d + 32) B x updates / HIP-event time of the segments (proposal launch + per iteration "
                                        "[torch callback, accept + next proposal]); split_design_* counts what this design must move per update: "
                                        "state or proposal in, proposal out, the callback's read of it, the accepted share written back"})
    # HBM traffic of the dominant kernel comes from separate rocprofv3 PMC passes (tools/gpu_profile.sh); the committed
    # summary is per launch of 100 steps on one named workload
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        key = "ndim=%d ntemps=%d nwalkers=%d mix=%s logl=%s" % (d, nt, W, a.mix, a.logl)
        if a.callback:                          # a fused-kernel profile says nothing about the split path: its own file, below
            break
        if tr.get("workload", "").startswith(key) and (tr.get("pick", "chain") == a.pick):
            per_launch = tr["traffic_bytes_per_launch"] * avg_steps / float(tr.get("steps_per_launch", 100))
            out["roofline"]["traffic"] = per_launch
            out["roofline"]["traffic_note"] = ("HBM bytes per launch from rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, "
                                               + ", ".join(tr["source"]))
            out["roofline"]["algorithmic_bytes_per_launch"] = bytes_per_update * upd_per_launch
            break
    if a.callback:
        # counter traffic of the split path (tools/gpu_profile.sh callback: separate rocprofv3 --pmc passes over this very command), per
        # iteration of every chain
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r06_callback_traffic.json")))[a.callback_kind]
            if tr.get("launches") == a.callback_launches and tr.get("workload", "").startswith("ndim=%d ntemps=%d nwalkers=%d" % (d, nt, W)):
                out["roofline"]["traffic"] = tr["traffic_bytes_per_iteration"] * avg_steps
                out["roofline"]["traffic_note"] = "HBM bytes per segment from rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, " + ", ".join(tr["source"])
                out["roofline"]["traffic_over_model"] = tr["traffic_bytes_per_iteration"] / (out["roofline"]["split_design_bytes_per_update"] * nt * W)
                out["roofline"]["traffic_gbs"] = tr["traffic_bytes_per_iteration"] * avg_steps / (avg_launch_ms * 1e-3) / 1e9
        except (OSError, ValueError, KeyError):
            pass
    # What the launch MUST move, counted live from this run's own state (not a counter): every chain's row, lnL and lp in and out,
    # and the rank-0 rows and flag words it stored (am_mode rle: only the accepted steps' rows -- the stored share is read off the
    # AM flags of the ring as it stands after the run).  The PMC figure above is a committed profile of the same workload; a
    # kernel change that adds traffic moves one of the two, and their ratio says so.
    if not a.callback and a.mix != "nuts" and getattr(eng, "owns_cold", False) and hasattr(eng, "t"):
        stored = stored_timed if stored_timed is not None else 1.0
        per_launch_model = 2.0 * nt * W * (8 * d + 16) + W * avg_steps * (stored * 8 * d + (8 if getattr(eng, "am_rle", False) else 0))
        out["roofline"]["traffic_model_bytes"] = per_launch_model
        out["roofline"]["am_rows_stored_share"] = stored
        if out["roofline"].get("traffic"):
            out["roofline"]["traffic_over_model"] = out["roofline"]["traffic"] / per_launch_model
    # cycles with AM entries pay 2 d^2 flop per AM proposal on the matrix cores ("+2d^2 per AM proposal"), whatever the
    # likelihood: the share of AM picks is w_am / sum(w) over the entries that are in the cycle during the timed region (DE joins
    # it after `burn` iterations)
    w_on = [weights[0], weights[1], weights[2] if getattr(eng, "de_on", False) else 0]
    f_am = (w_on[1] / float(sum(w_on))) if (len(weights) == 3 and sum(w_on) > 0 and a.mix != "nuts") else 0.0
    if a.logl == "dense" or f_am > 0:
        # config 3: the dense contraction bounds the kernel: 2d^2+3d flop per likelihood, +2d^2 per AM proposal.
        # The likelihood is summed over half of the symmetric precision matrix: d^2 + 3d flop executed per evaluation (SURVEY's
        # 2d^2 + 3d prices the full product)
        base = (d * d + 3 * d) if a.logl == "dense" else flops_per_update
        flops = base + 2 * d * d * f_am
        tfd = flops * upd_per_launch / (avg_launch_ms * 1e-3) / 1e12
        out["roofline"].update({"bound": "mfma", "achieved": tfd, "frac": tfd / F64_PEAK_TFLOPS, "algorithmic_flops_per_update": flops,
                                "am_pick_share": f_am, "frac_of_sustained_mfma": tfd / F64_MFMA_SUSTAINED,
                                "flops_note": ("executed flops per update: %s + 2d^2 x the share of AM picks (%.3f)" % (
                                    "the quadratic form over half of the symmetric precision matrix (d^2 + 3d)" if a.logl == "dense"
                                    else "4d (proposal + isotropic likelihood)", f_am))})
        if a.logl == "dense":
            out["roofline"]["flops_note"] += "; the full-matrix count 2d^2 + 3d would read %.3f of peak" % (
                (flops + d * d) * upd_per_launch / (avg_launch_ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS)
    if rank == 0:
        acc = eng.get("nacc").astype(np.float64)
        out["acceptance_rank0_mean"] = float(acc[:, 0].mean() / max(1, eng.iter))        # over the whole run, ESS leg included
        out["swap_accept_rate_pair0"] = float(eng.get("nswap")[:, 0].mean() / max(1, eng.swap_proposed))
        out.update(ess_out)
    del eng
    import gc
    gc.collect()                            # the legs behind the headline build their engines in this process: give the memory back
    torch.cuda.empty_cache()
    return out


# The other BASELINE configs as short legs behind the headline (one GPU): name -> (argument overrides, steps, warmup).  The default
# mixes start their DE proposals after burn = 10000 iterations = 100 steps, hence their warmup.
ALSO = (
    ("config3_dense_scam", dict(logl="dense"), 20, 10),
    ("config2_default_mix", dict(mix="default"), 20, 105),
    ("config3_dense_default_mix_walker_pick", dict(logl="dense", mix="default", pick="walker"), 12, 105),
    ("config2_replica_per_walker_cov_device_ql", dict(cov_mode="per_walker_device"), 20, 10),
    ("config4_share_1000d_64x512", dict(ndim=1000, nwalkers=512), 30, 20),
    ("config5_share_curved_nuts_16x4096", dict(logl="curved", ndim=20, ntemps=16, mix="nuts"), 6, 4),
    # the headline's workload with the likelihood OUTSIDE the library: a batched torch callback on the device tensor of proposals
    # (the reference's logl / logp boundary, PTMCMCSampler.py:605-611, 1072-1086)
    ("config2_batched_callback", dict(callback=True, callback_kind="hip"), 20, 5),                 # the callback is a HIP kernel behind the C ABI
    ("config2_batched_callback_torch", dict(callback=True, callback_kind="norm"), 20, 5),         # the callback is a torch expression
    ("config2_batched_callback_default_mix", dict(callback=True, callback_kind="hip", mix="default"), 10, 105),   # AM increments from the matrix cores ahead of the proposals
)
L2_PEAK_TBS = 34.5             # MI355X_MICROARCH.md: aggregate L2 bandwidth (4 MiB per XCD); profiles/r05_row_gather.txt: random 8 KB rows of a
                               # 7.8 MB table (it does not fit one XCD's L2: half the rows come from the MALL) arrive at 16.1 TB/s


def also_legs(a, rank, world, local, dist, backend):
    import copy
    import torch
    res = {}
    if a.preheat > 0:                           # bring the clocks up once (bench.py --preheat), the legs then run back to back
        hm = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
        t_h = time.perf_counter()
        while time.perf_counter() - t_h < a.preheat:
            for _ in range(8):
                hm @ hm
            torch.cuda.synchronize()
        del hm
    t_all = time.perf_counter()
    for name, over, steps, warmup in ALSO:
        b = copy.copy(a)
        b.__dict__.update(over)
        b.steps, b.warmup, b.ess_window, b.weights, b.preheat = steps, warmup, 0, None, 0.0     # the clocks are up: the headline just ran
        t0 = time.perf_counter()
        try:
            o = measure(b, rank, world, local, dist, backend)
        except Exception as e:              # noqa: BLE001 -- a leg that fails must not take the headline's line with it
            res[name] = {"error": repr(e)[:300]}
            log("also %s FAILED: %r" % (name, e))
            continue
        r = o["roofline"]
        leg = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": steps, "warmup": warmup,
               "workload": o["config"]["workload"], "iterations_timed": o["iterations_timed"], "cov_epochs_timed": o["cov_epochs_timed"],
               "acceptance_rank0_mean": o.get("acceptance_rank0_mean"),
               "roofline": {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms", "steps_per_launch",
                                                  "kernel_time_share_of_wall", "algorithmic_flops_per_update", "algorithmic_bytes_per_update",
                                                  "split_design_bytes_per_update", "split_design_gbs", "acceptance_whole_run", "traffic",
                                                  "traffic_over_model", "traffic_gbs", "f64_valu_frac", "launch_ms_min", "launch_ms_max", "leapfrogs_timed",
                                                  "nuts_calls_timed", "leapfrogs_per_nuts_call", "flops_per_leapfrog", "flops_note", "frac_of_sustained_mfma") if k in r},
               "leg_seconds": time.perf_counter() - t0}
        if b.ndim > 416 and b.mix == "scam":
            # 64 lanes per chain: a step reads ONE table row of 8 * ndim bytes per chain from L2 / MALL (the table, 8 MB at ndim = 1000,
            # does not fit an XCD's 4 MB L2); that gather bounds the kernel, not the 4d flop of the row arithmetic
            tbs = 8.0 * b.ndim * b.ntemps * b.nwalkers * r["steps_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12
            leg["roofline"].update({"f64_valu_frac": r["frac"], "bound": "l2", "achieved": tbs, "peak": L2_PEAK_TBS, "unit": "TB/s",
                                    "frac": tbs / L2_PEAK_TBS, "gather_floor_tbs": 16.1,
                                    "note": "table rows read per launch / launch time against the aggregate L2 peak; a bare gather of random "
                                            "8 KB rows of a table this size runs at 16.1 TB/s (tools/row_gather_bw.hip, profiles/r05_row_gather.txt)"})
        res[name] = leg
        log("also %-44s %.4g upd/s  launch %.3f ms  step %.3f ms  (%.1f s)" % (name, leg["value"], r["avg_launch_ms"], leg["ms_per_step"], leg["leg_seconds"]))
    res["total_seconds"] = time.perf_counter() - t_all
    return res


def sharded_selfcheck(rank, world, local, dist):
    """Pre-flight of `--gpus N` (temperature blocks), BEFORE the warmup and outside the timed region: a small ladder -- 16 ranks per GPU
    x 64 walkers x 100-d, default SCAM / AM / DE cycle, pooled covariance, covUpdate 100, burn 200, Tskip 10, 300 iterations: three
    covariance epochs, a DE epoch and DE activation, 30 swap epochs whose rows cross the block edges -- runs (a) sharded over the
    backend the bench is about to time (RCCL: lnL gather, neighbour send/recv, table and DE-row broadcasts) and (b) as ONE engine on
    this rank's own GPU; every rank compares ITS block of (a) with (b) bit for bit (states, likelihoods, counters, table).  The
    number the bench prints then comes with its own proof that the multi-GPU path computes what the one-GPU path does
    (PTMCMCSampler.py:631-697, 545-576)."""
    import numpy as np
    import torch
    from ptmcmcsampler_amd.engine import PTEngine
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    d, ntb, W, n = 100, 16, 64, 300
    ntg = ntb * world
    kw = dict(weights=(20, 20, 20), cov_update=100, burn=200, tskip=10, seed=777, cov_mode="pooled", device=local)
    cov0 = np.eye(d) * 0.01
    p0 = np.random.RandomState(5).randn(W, ntg, d) * 0.1
    t0 = time.perf_counter()
    s = ShardedPTEngine(d, ntg, W, cov0, group=dist.group.WORLD, **kw)
    s.init_state(p0)
    s.run(n)
    s.sync()
    g = PTEngine(d, ntg, W, cov0, **kw)
    g.init_state(p0)
    g.run(n)
    g.sync()
    L, sl = s.local, slice(rank * ntb, (rank + 1) * ntb)
    bad = []
    for name in ("X", "lnL", "lp"):
        if not np.array_equal(L.by_temp(name), g.by_temp(name)[:, sl]):
            bad.append(name)
    for name in ("nacc", "jstat"):
        if not np.array_equal(L.get(name), g.get(name)[:, sl]):
            bad.append(name)
    if not np.array_equal(L.get("nswap")[:, sl], g.get("nswap")[:, sl]):
        bad.append("nswap")
    for name in ("Ut", "S"):
        if not np.array_equal(L.get(name), g.get(name)):
            bad.append(name)
    if L.exchange_violations() != 0:
        bad.append("exchange_violations")
    # rows did cross this rank's upper block edge (accepted swaps of the pair that straddles it), and some state is not where it started
    edge = int(g.get("nswap")[:, (rank + 1) * ntb - 1].sum()) if rank + 1 < world else int(g.get("nswap")[:, rank * ntb - 1].sum())
    allf = torch.zeros((world, 3), dtype=torch.float64, device="cuda")          # every rank fills its row; one all-reduce (as the wall time's)
    allf[rank] = torch.tensor([0.0 if bad else 1.0, float(edge), float(s.neighbour_swaps)], dtype=torch.float64)
    dist.all_reduce(allf, op=dist.ReduceOp.SUM)
    allf = allf.cpu().numpy().astype(np.int64)
    ok = bool(allf[:, 0].min() == 1 and allf[:, 1].min() > 0 and s.swap_proposed == n // 10)
    if bad:
        log("sharded selfcheck: rank %d differs in %s" % (rank, bad))
    del s, g
    torch.cuda.empty_cache()
    return ok, {"ranks_per_gpu": ntb, "nwalkers": W, "ndim": d, "iterations": n, "swap_epochs": n // 10,
                "neighbour_swaps": int(allf[0, 2]), "edge_swaps_accepted_min_over_edges": int(allf[:, 1].min()),
                "blocks_equal_single_engine": [bool(v) for v in allf[:, 0]], "seconds": time.perf_counter() - t0,
                "what": "a %d-rank ladder sharded over the %d ranks of this run vs ONE engine on each rank's own GPU: every rank's block bit for bit "
                        "(X, lnL, lp, nacc, jstat, nswap, Ut, S), before the warmup, outside the timed region" % (ntg, world)}


def main():
    a = parse()
    if a.also == "auto":
        dflt = parse_defaults()
        a.also = all(getattr(a, k) == getattr(dflt, k) for k in ("ndim", "ntemps", "nwalkers", "mix", "weights", "pick", "logl", "prior", "cov_mode",
                                                                 "swap_mode", "partition", "sharded", "callback", "callback_kind", "callback_launches", "callback_graph", "am_mode", "eig_lag", "stats_async")) and a.gpus == 1
    else:
        a.also = a.also == "on"
    if a.gpus > 1 and "LOCAL_RANK" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    # stdout must carry exactly one JSON line: park everything else the process (and RCCL's C-level banner)
    # prints on stderr, and keep the real stdout for the result
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (a.gpus, world))
    weights = cycle_weights(a)
    cpu = None
    # the CPU baseline is timed on rank 0 of the single-GPU run only; the NumPy port covers the Gaussian configs
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.mix != "nuts" and a.logl == "iso":
        cpu = cpu_baseline(a, weights)
        log("cpu baseline %.3g updates/s on %d cores" % (cpu["value"], cpu["cores"]))
    import numpy as np
    import torch
    backend = os.environ.get("PTMI_DIST_BACKEND", "nccl")          # "gloo" only to rehearse N ranks on a one-GPU box
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("--gpus %d but only %d GPUs are visible (PTMI_DIST_BACKEND=gloo rehearses several ranks on one GPU)"
                         % (world, torch.cuda.device_count()))
    local %= max(1, torch.cuda.device_count())
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

    if a.also_child:                            # the legs of the other configs, in a process of their own (see below)
        os.write(real_stdout, (json.dumps(also_legs(a, rank, world, local, dist, backend)) + "\n").encode())
        return
    check = None
    if dist is not None and a.partition == "temps" and world > 1:
        try:
            check = sharded_selfcheck(rank, world, local, dist)
            log("sharded selfcheck %s (%.1f s)" % ("ok" if check[0] else "FAILED", check[1]["seconds"]))
        except Exception as e:              # noqa: BLE001 -- the line must still come out, and say that its proof did not
            check = (False, {"error": repr(e)[:300]})
            log("sharded selfcheck raised: %r" % (e,))
    out = measure(a, rank, world, local, dist, backend)
    if check is not None:
        out["sharded_selfcheck"], out["sharded_selfcheck_detail"] = check
    if rank == 0 and world == 1 and a.also:
        # In a process of their own: a fault in one of the other configs' kernels must not take the headline's line with it.
        # (The headline's engine is gone and its memory returned; the child brings the clocks up itself.)
        cmd = [sys.executable, os.path.abspath(__file__), "--also-child", "--also", "off", "--no-cpu-baseline", "--ess-window", "0", "--preheat", "0.2"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["also"] = json.loads(lines[-1]) if lines else {"error": "no output (rc %d): %s" % (r.returncode, r.stderr[-300:])}
            sys.stderr.write(r.stderr[-4000:])
        except Exception as e:                  # noqa: BLE001
            out["also"] = {"error": repr(e)[:300]}
    if rank == 0:
        d, weights = a.ndim, cycle_weights(a)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            # the C oracle on one core, for scale (a compiled scalar port; not what a reference user gets)
            from oracle import oracle as orc
            o = orc.OracleEngine(d, 8, 8, np.eye(d) * 0.01, weights=weights, cov_update=1000, burn=10000, tskip=100,
                                 seed=1234, cov_mode="pooled" if a.cov_mode == "pooled" else "per_walker")
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
