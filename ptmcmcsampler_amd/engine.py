"""Batched parallel-tempering engine: host-side driver of libptmi.so.

One ``PTEngine`` owns the chains of ``nwalkers`` independent replicas x ``ntemps``
temperature ranks on one MI355X and advances them with the fused HIP kernels.  The
host part below is the epoch logic of the reference's ``PTMCMCOneStep``
(PTMCMCSampler/PTMCMCSampler.py:545-585: covariance epoch, DE epoch, DE activation)
and the iteration loop of ``sample()`` (:495-528), restated for a batch; everything
per-chain runs on the device.  Device memory is held in torch tensors (plumbing only:
allocation, host<->device copies, and RCCL in ``sharded.py``).
"""
import ctypes as C

import os

import numpy as np

from . import _lib
from .ladder import temperature_ladder


def _torch():
    import torch
    return torch


_BLAS_CTL = None


def _blas_threads(n):
    """Context that caps the BLAS thread pools at ``n``.  The controller is built once: discovering the loaded BLAS
    libraries is what costs (0.2 ms per call, a third of a 100 x 100 factorization)."""
    global _BLAS_CTL
    try:
        if _BLAS_CTL is None:
            from threadpoolctl import ThreadpoolController
            _BLAS_CTL = ThreadpoolController()
        return _BLAS_CTL.limit(limits=n)
    except ImportError:
        import contextlib
        return contextlib.nullcontext()


def _blas_single_thread():
    return _blas_threads(1)


def _usable_cores():
    """Visible CPUs, capped by the affinity mask and the cgroup CPU quota."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def factorize(cov, per_walker):
    """Eigen-directions U (columns) and variances S of a jump covariance.

    Per-walker mode is the reference's call, ``np.linalg.svd`` (PTMCMCSampler.py:145, 803), on one BLAS thread (LAPACK's
    last bits depend on the thread count; a 100 x 100 factorization gains nothing from more).  Pooled mode is not a
    replica of a reference run and uses the symmetric solver: ``np.linalg.eigh``, eigenvalues by decreasing size and
    in absolute value (what the SVD of a symmetric matrix returns up to signs), 2.5x cheaper at 1000 x 1000 where the
    factorization dominates the covariance epoch; beyond 256 parameters it runs on 8 BLAS threads."""
    if per_walker:
        with _blas_threads(1):
            U, S, _ = np.linalg.svd(cov)
        return U, S
    with _blas_threads(1 if len(cov) <= 256 else 8):
        w, V = np.linalg.eigh(cov)
    return np.ascontiguousarray(V[:, ::-1]), np.abs(w[::-1])



def interval_par(a, b, d):
    """Parameters of the ("interval", a, b) family (include/ptmi.h PTMI_LOGL_INTERVAL): a, w = b - a, log w."""
    a = np.broadcast_to(np.asarray(a, np.float64), (d,))
    b = np.broadcast_to(np.asarray(b, np.float64), (d,))
    if not np.all(b > a):
        raise ValueError("interval logl needs a < b")
    w = b - a
    return np.ascontiguousarray(np.concatenate([a, w, np.log(w)]))


class PTEngine(object):
    """Chains of ``nwalkers`` x ``ntemps`` on one GPU.

    Parameters mirror the reference's sampler: ``cov0`` is the initial jump covariance
    (``PTSampler.__init__`` ``cov``), ``weights`` = (SCAMweight, AMweight, DEweight),
    ``cov_update`` = covUpdate, ``burn`` = burn (also the DE-buffer length), ``tskip`` =
    Tskip.  ``cov_mode``: ``"per_walker"`` makes every walker a faithful replica of a
    reference run (own covariance, eigenvectors and DE history); ``"pooled"`` adapts one
    covariance from all walkers' rank-0 samples.  ``logl`` / ``logp`` select the built-in
    device likelihood / prior: ("iso",), ("dense", mu, P) with P a (symmetric) precision matrix, ("curved",),
    ("interval", a, b) = the unit Gaussian on the box (a, b) in the coordinates of the reference's ``intervalTransform`` (its own NUTS
    workload, tests/test_nuts.py:13-140; ndim <= 512); ("flat",), ("box", lo, hi).  ``swap_mode``: ``"sweep"`` is the reference's hot -> cold PTswap; ``"oddeven"`` tries
    the disjoint pairs (k, k+1), k = swap epoch (mod 2), all at once (see include/ptmi.h).  ``pick_mode``: ``"chain"`` =
    every chain draws its own entry of the proposal cycle (the reference's ``_jump``); ``"walker"`` = one draw per walker
    and iteration fixes the proposal type of all its temperature ranks (wave-uniform on the device, include/ptmi.h).
    ``nuts_maxdepth``: NUTS stops doubling at this tree height.  The reference doubles ``while s == 1`` with no cap
    (nutsjump.py:716-802); the default 24 (2^24 leapfrogs in one call, the ABI's limit) is out of reach of any run, i.e. the
    reference's behaviour.  A lower value is an engine option (bounded cost per call).  Memory: the tree scratch holds
    7 + 4 (maxdepth + 1) vectors of ndim doubles and 4 (maxdepth + 1) scalars per chain (107 vectors at 24, 51 at 10):
    86 KB per chain at ndim = 100, i.e. 22 GB for 262 144 chains -- lower it for very large batches.
    ``eig_mode``: who factorizes the adapted covariance at a covariance epoch (PTMCMCSampler.py:797-803): ``"lapack"`` = the
    host, exactly as the reference (``np.linalg.svd`` per walker); ``"jacobi"`` = ``ptmi_eig_jacobi`` on the device, one
    block per walker, no host round trip (ndim <= 101, one parameter group; same subspaces, its own sign rule); ``"ql"`` =
    ``ptmi_eig_ql``, Householder tridiagonalization + implicit QL on the device (ndim <= 128; a quarter of the Jacobi kernel's time
    on nearly degenerate spectra, two matrices per CU: the choice for thousands of per-walker covariances);
    ``"hipsolver"`` = the ROCm library's symmetric eigensolver on the engine's stream (``torch.linalg.eigh`` on the device
    tensor: no host round trip either; with parameter groups one call per group's block) -- the choice for large ndim, where the host's LAPACK call is the
    epoch (1000 x 1000: 22 ms against 83 ms on 8 host threads; at ndim = 100 the host's 0.6 ms wins).  Like LAPACK's, its
    last bits are the library's: such a run is not bit-reproducible against the oracle, only its decomposition is checked.
    ``"sytrd"`` = ``ptmi_eig_sytrd`` for ONE large pooled covariance (ndim <= 1024), all of it the library's own kernels: Householder
    tridiagonalization in one kernel with the matrix in the LDS of 64 blocks (a quarter of the CUs), the tridiagonal matrix's
    eigenvectors by divide and conquer (csrc/ptmi_dc.inc.h), back-transformed through the reflectors -- 12.6 ms at 1000 x 1000 where the
    ROCm library's own reduction is 7000 launches of one-block kernels, two thirds of its 35 ms.  No oracle restates its last bits; the
    tests check every table it makes against the covariance and step the oracle's chains with it.
    ``eig_lag`` (L >= 0 launches; pooled covariance): the eigenvectors of a covariance epoch take effect L launches late --
    ``run`` queues the L launches that follow the epoch (and their swaps) with the table in force, the factorization runs
    MEANWHILE, and the launch after them uses the result (a new epoch finishes a pending one first).  The reference applies the
    table at once (PTMCMCSampler.py:560); the pooled covariance is an engine mode anyway, and with L = 1 the adaptation sees its
    table a tenth of a period late.  ``eig_mode="lapack"``: the host factorizes while the GPU samples (L = 1 hides it at
    ndim = 100); ``eig_mode="hipsolver"``: the library's kernels run on a side stream BESIDE the launches (22 ms at ndim = 1000
    beside 2.6 ms launches: L = 9).  Same statistics, same factorization, no GPU idle time at the epoch (oracle:
    ``OracleEngine(eig_lag=L)``).  ``eig_mode="ql"`` (pooled or per-walker covariances, one GPU): ``ptmi_eig_ql_from`` on the side
    stream; built and bit-exact, but no faster at 4096 walkers -- the step launches slow down by what the factorization takes.
    ``am_mode``: how the rank-0 chain's samples (updateChains' buffer, PTMCMCSampler.py:327-328) are kept between covariance
    epochs.  ``"rows"``: every step stores its row.  ``"rle"`` (pooled covariance): a rejected proposal leaves the chain where it
    was, so a step stores its row only when it was accepted (or is a KEY row: first step of a launch, ring rows 0 and 1, the
    swap's row) plus a flag word, and the pooled statistics take every stored row once, weighted by its run length (include/ptmi.h
    ``AMflag``; oracle: ``orc_pool_update_rle`` -- the same sample covariance, summed in another order).  Readers that want
    every row (``get("AM")``, the DE history, chain files) get the repeats copied forward first (``am_expand``).
    ``"auto"`` = ``"rle"`` where it applies.
    ``stats_async`` (pooled covariance with ``eig_lag >= 1``): the statistics of a covariance period that is over need nothing the next
    launches touch once those write ANOTHER ring -- so the engine keeps two rings (``t["AM"]`` is always the one in use), switches at
    every covariance epoch, and runs the period's statistics (``ptmi_update_cov_on``) and the factorization behind them on a side
    stream BESIDE the launches that follow; the table still takes effect ``eig_lag`` launches after the epoch.  A scheduling change
    only: every result equals the run without it bit for bit (oracle: ``OracleEngine(eig_lag=L)``).  Needs burn to be a multiple of
    cov_update when a DE history is kept (a DE epoch then reads the finished period's ring before the switch).
    """

    def __init__(self, ndim, ntemps, nwalkers, cov0, ladder=None, logl=("iso",), logp=("flat",),
                 weights=(20, 20, 20), cov_update=1000, burn=10000, tskip=100, seed=0,
                 cov_mode="per_walker", hot_chain=False, Tmin=1, Tmax=None,
                 ntemps_global=None, temp0=0, walker0=0, device=0, split=False, use_de_buffer=None,
                 w_host=0, keep_lnl=False, groups=None, swap_mode="sweep",
                 grad_weights=(0, 0), hmc=(0.1, 2, 300), nuts_delta=0.6, nuts_maxdepth=24, pick_mode="chain",
                 eig_mode="lapack", am_mode="auto", eig_lag=0, stats_async=False):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available() or _lib.device_count() < 1:
            raise _lib.PtmiError("no MI355X visible: the engine has no CPU fallback")
        self.d, self.nt, self.W = int(ndim), int(ntemps), int(nwalkers)
        self.ntg = self.nt if ntemps_global is None else int(ntemps_global)
        self.temp0, self.walker0 = int(temp0), int(walker0)
        self.ladder = np.ascontiguousarray(
            temperature_ladder(self.ntg, ndim, Tmin, Tmax) if ladder is None else ladder, dtype=np.float64)
        if len(self.ladder) != self.ntg:
            raise ValueError("ladder has %d entries for %d temperatures" % (len(self.ladder), self.ntg))
        self.temps_mh = self.ladder[self.temp0:self.temp0 + self.nt].copy()
        if hot_chain and self.temp0 + self.nt == self.ntg:
            self.temps_mh[-1] = 1e80                                  # PTMCMCSampler.py:281-282
        self.cov_update, self.burn, self.tskip, self.seed = int(cov_update), int(burn), int(tskip), int(seed)
        self.weights = tuple(int(w) for w in weights)
        self.per_walker = cov_mode == "per_walker"
        if cov_mode not in ("per_walker", "pooled"):
            raise ValueError("cov_mode must be 'per_walker' or 'pooled'")
        if pick_mode not in _lib.PICK_MODES:
            raise ValueError("pick_mode must be 'chain' or 'walker'")
        self.pick_mode = pick_mode
        if eig_mode not in ("lapack", "jacobi", "ql", "hipsolver", "sytrd"):
            raise ValueError("eig_mode must be 'lapack', 'jacobi', 'ql', 'hipsolver' or 'sytrd'")
        if eig_mode == "sytrd" and (cov_mode != "pooled" or not 3 <= int(ndim) <= 1024):
            raise ValueError("eig_mode='sytrd' factorizes one pooled covariance of 3 <= ndim <= 1024")
        self.eig_mode = eig_mode
        if int(eig_lag) < 0:
            raise ValueError("eig_lag must be >= 0 launches")
        # (where the late table is not implemented -- per-walker covariances unless eig_mode is "ql", parameter groups, the on-stream
        # device eigensolver "jacobi" -- the table is applied at once, as OracleEngine defines it too)
        if int(eig_lag) > 0 and eig_mode == "jacobi" and cov_mode == "pooled" and groups is None:
            raise ValueError("eig_lag > 0 is implemented for eig_mode 'lapack', 'hipsolver', 'sytrd' and 'ql' (got %r)" % (eig_mode,))
        lag_ok = groups is None and ((cov_mode == "pooled" and eig_mode in ("lapack", "hipsolver", "sytrd")) or eig_mode == "ql")
        self.eig_lag, self._eig_pending, self._eig_wait = (int(eig_lag) if lag_ok else 0), False, 0
        self.Wc = self.W if self.per_walker else 1
        # parameter groups (PTMCMCSampler.py:129-145): per-group eigenvectors, embedded in the full space
        self.groups = [np.arange(self.d)] if groups is None else [np.asarray(g, dtype=np.int64) for g in groups]
        self.ngr = len(self.groups)
        # "whole": one group that IS the full parameter vector in order (a permutation of it needs put_eig's embedding)
        self.whole = self.ngr == 1 and np.array_equal(self.groups[0], np.arange(self.d))
        if eig_mode in ("jacobi", "sytrd") and not self.whole:
            raise ValueError("eig_mode=%r factorizes the full covariance: no parameter groups (eig_mode='ql' and 'hipsolver' take them)" % eig_mode)
        if eig_mode == "ql" and not self.whole and any(len(np.unique(g)) != len(g) for g in self.groups):
            raise ValueError("eig_mode='ql' with parameter groups: a group may not repeat a parameter")
        self.gsize = np.ascontiguousarray([len(g) for g in self.groups], dtype=np.int32)
        self.gmask = np.zeros((self.ngr, self.d))
        for gi, g in enumerate(self.groups):
            if len(g) < 1 or g.min() < 0 or g.max() >= self.d:
                raise ValueError("group %d has indices outside [0, %d)" % (gi, self.d))
            self.gmask[gi, g] = 1.0
        # gradient jumps on the built-in likelihoods (PTMCMCSampler.py:225-258): (NUTSweight, HMCweight); the whitening
        # comes from the INITIAL covariance and is never adapted (nutsjump.py:45, 53-54)
        self.grad_weights = tuple(int(w) for w in grad_weights)
        has_gj = sum(self.grad_weights) > 0
        self.gj_tab = np.zeros(0)
        if has_gj:
            import scipy.linalg as sl
            L = sl.cholesky(np.asarray(cov0, dtype=np.float64), lower=True)
            Li = sl.solve_triangular(L, np.eye(self.d), trans=0, lower=True)
            self.gj_tab = np.ascontiguousarray(np.stack([L, Li, L.T]))
        self.device = torch.device("cuda", device)
        self.dev_index = device
        d, nt, W, Wc = self.d, self.nt, self.W, self.Wc
        f64, i32, i64 = torch.float64, torch.int32, torch.int64
        z = lambda shape, dt=f64: torch.zeros(shape, dtype=dt, device=self.device)  # noqa: E731
        has_de = self.weights[2] > 0 if use_de_buffer is None else use_de_buffer
        st, ep = C.c_int(0), C.c_int(0)                                # row format of the DE buffer (include/ptmi.h)
        gshape = has_gj or logl[0] == "interval"                       # the interval family lives in the gradient-jump shapes (include/ptmi.h)
        _lib.check(self.lib.ptmi_de_row_stride(d, int(gshape), C.byref(st), C.byref(ep)))
        self.de_ld, self.de_epl = st.value, ep.value
        _lib.check(self.lib.ptmi_am_row_format(d, int(gshape), C.byref(ep)))         # row format of the AM buffer (include/ptmi.h)
        self.am_epl = ep.value
        self.am_pos = self.am_inv = None
        if self.am_epl:
            e_, ln_ = np.arange(d) // 4, np.arange(d) % 4
            self.am_pos = np.where(e_ < 2 * (self.am_epl // 2), 8 * (e_ // 2) + 2 * ln_ + e_ % 2, 8 * (self.am_epl // 2) + ln_)   # where parameter i sits
            self.am_inv = np.argsort(self.am_pos)                                   # which parameter sits at position p
        self.owns_cold = self.temp0 == 0
        if am_mode not in ("auto", "rows", "rle"):
            raise ValueError("am_mode must be 'auto', 'rows' or 'rle'")
        rle_ok = not self.per_walker and self.owns_cold                  # = ptmi_am_flags_ok
        if am_mode == "rle" and not rle_ok:
            raise ValueError("am_mode='rle' serves the pooled covariance (cov_mode='pooled')")
        self.am_rle = rle_ok and am_mode != "rows"
        self.t = dict(
            X=z((W, nt, d)), lnL=z((W, nt)), lp=z((W, nt)),
            temp_of=torch.arange(nt, dtype=i32, device=self.device).repeat(W, 1).contiguous(),
            slot_of=torch.arange(nt, dtype=i32, device=self.device).repeat(W, 1).contiguous(),
            Ut=z((Wc, self.ngr, d, d)), S=z((Wc, self.ngr, d)),
            DE=z((Wc, self.burn, self.de_ld)) if has_de else None,
            AM=z((W, self.cov_update, d)) if self.owns_cold else None,
            nacc=z((W, nt), i64), jstat=z((W, nt, _lib.J_NTYPES, 2), i64), nswap=z((W, self.ntg), i64),
            mu=z((Wc, d)) if self.owns_cold else None, M2=z((Wc, d, d)) if self.owns_cold else None,   # pooled: ONE running (mu, M2)
            cov=z((Wc, d, d)),
            Q=z((W, nt, d)) if split else None, qaux=z((W, nt, 4)) if split else None,
            # the second proposal buffer and the chains' state locations (include/ptmi.h: accepted proposals stay where they are)
            Q2=z((W, nt, d)) if split else None, sloc=z((W, nt), i32) if split else None,
            AMaux=z((W, self.cov_update, 2)) if (keep_lnl and self.owns_cold) else None,
            gj=z((W, nt, _lib.GJ_NSTATE)) if has_gj else None,
            AMflag=z((W, self.cov_update), i64) if self.am_rle else None,     # AM row flags (include/ptmi.h): every row starts as KEY
        )
        if self.am_rle:
            self.t["AMflag"].fill_(_lib.AMROW_KEY)
        if has_gj:
            self.t["gj"][..., _lib.GJ_EPSBAR] = 1.0
        cov0 = np.asarray(cov0, dtype=np.float64)
        self.t["cov"].copy_(torch.from_numpy(np.broadcast_to(cov0, (Wc, d, d)).copy()))
        # likelihood / prior parameters
        self._par_l = np.zeros(0)
        self._par_p = np.zeros(0)
        if logl[0] == "dense":
            mu, P = np.asarray(logl[1], np.float64), np.asarray(logl[2], np.float64)
            self._par_l = np.concatenate([mu, np.ascontiguousarray(P.T).ravel()])
        if logl[0] == "interval":
            self._par_l = interval_par(logl[1], logl[2], d)
        if logp[0] == "box":
            self._par_p = np.concatenate([np.asarray(logp[1], np.float64), np.asarray(logp[2], np.float64)])
        self.stream = torch.cuda.current_stream(self.device)
        cfg = _lib.Config(
            ndim=d, ntemps=nt, nwalkers=W, ntemps_global=self.ntg, temp0=self.temp0, walker0=self.walker0,
            logl_kind=_lib.LOGL[logl[0]], logp_kind=_lib.LOGP[logp[0]], w_host=int(w_host), w_scam=self.weights[0], w_am=self.weights[1],
            w_de=self.weights[2] if has_de else 0, de_size=self.burn, cov_update=self.cov_update, tskip=self.tskip,
            cov_per_walker=int(self.per_walker), device=device, ngroups=self.ngr if self.ngr > 1 else 0,
            swap_mode=_lib.SWAP_MODES[swap_mode], pick_mode=_lib.PICK_MODES[pick_mode], seed=self.seed,
            w_nuts=self.grad_weights[0], w_hmc=self.grad_weights[1], gj_nburn=self.burn, hmc_eps=float(hmc[0]),
            hmc_min=int(hmc[1]), hmc_max=int(hmc[2]), nuts_maxdepth=int(nuts_maxdepth), nuts_delta=float(nuts_delta),
            gj_tab=self.gj_tab.ctypes.data_as(_lib._dp) if has_gj else None,
            group_size=self.gsize.ctypes.data_as(C.POINTER(C.c_int32)), group_mask=self.gmask.ctypes.data_as(_lib._dp),
            stream=C.c_void_p(self.stream.cuda_stream),
            ladder=self.ladder.ctypes.data_as(_lib._dp), temps_mh=self.temps_mh.ctypes.data_as(_lib._dp),
            logl_par=self._par_l.ctypes.data_as(_lib._dp) if len(self._par_l) else None, logl_par_len=len(self._par_l),
            logp_par=self._par_p.ctypes.data_as(_lib._dp) if len(self._par_p) else None, logp_par_len=len(self._par_p))
        buf = _lib.Buffers(**{k: (C.c_void_p(v.data_ptr()) if v is not None else None) for k, v in self.t.items()})
        self.h = C.c_void_p()
        _lib.check(self.lib.ptmi_create(C.byref(cfg), C.byref(buf), C.byref(self.h)))
        self.de_on = False
        self.de_head = 0
        self.iter = 0
        self.swap_proposed = 0
        self.eig_epochs = 0
        # a pending device factorization is finished BEHIND the statistics of the next covariance epoch (update_cov); the same on
        # every block of a sharded ladder (from the configuration alone: ShardedPTEngine orders its broadcasts by it)
        self.late_finish = self.eig_lag > 0 and eig_mode in ("hipsolver", "sytrd", "ql") and not stats_async
        self.stats_async = False
        if stats_async:
            # from the configuration alone, on EVERY block of a sharded ladder: a block without rank 0 has no statistics to run, but it
            # must refuse what the owner refuses -- else the owner raises alone and the others wait in their first collective for ever
            ok = (not self.per_walker and self.eig_lag >= 1 and self.whole and eig_mode in ("lapack", "hipsolver", "sytrd")
                  and (not has_de or self.burn % self.cov_update == 0))
            if not ok:
                raise ValueError("stats_async needs a pooled covariance, eig_lag >= 1, eig_mode 'lapack', 'hipsolver' or 'sytrd', one "
                                 "parameter group, and burn a multiple of cov_update when a DE history is kept")
        if stats_async and self.owns_cold:                            # (a block without rank 0 has no statistics to run)
            self.stats_async = True
            # the second ring (and its flags): what the statistics of the period before read while this period is written
            self._alt = dict(AM=torch.zeros_like(self.t["AM"]),
                             AMaux=torch.zeros_like(self.t["AMaux"]) if self.t["AMaux"] is not None else None,
                             AMflag=torch.full_like(self.t["AMflag"], _lib.AMROW_KEY) if self.t["AMflag"] is not None else None)
            # high priority: the statistics' blocks go ahead of the step kernel's next blocks wherever a CU has room for them
            self._st = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("PTMI_STATS_PRIO", "-1")))     # (0 measured the same, round 5)
            self._ev_period, self._ev_stats = torch.cuda.Event(), torch.cuda.Event()
            self._stats_queued = False
        self._eig_host(0, cov0)                                       # every walker starts from the same covariance
        if Wc > 1:
            self.t["Ut"][1:] = self.t["Ut"][0]
            self.t["S"][1:] = self.t["S"][0]

    def __del__(self):
        try:
            if getattr(self, "_eig_pool", None) is not None:
                self._eig_pool.terminate()
                self._eig_pool = None
            if getattr(self, "h", None) is not None and self.h:
                self.lib.ptmi_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ data access
    def get(self, name):
        """Device array -> numpy (counters as uint64; DE rows in parameter order whatever their device format)."""
        if name == "AM":
            self.am_expand()
        if self.stats_async and name in ("cov", "mu", "M2"):
            self._st.synchronize()                                    # the statistics run beside the launches
        a = self.t[name].cpu().numpy()
        if self.stats_async and name in ("AM", "AMflag", "AMaux"):
            # ONE ring as a reader of the reference's buffer sees it: the rows of the current period come from the ring in use,
            # the older ones from the other ring
            lo, hi = self.am_period()
            cur = np.zeros(self.cov_update, dtype=bool)
            cur[np.arange(lo, hi + 1) % self.cov_update] = True
            b = self._alt[name].cpu().numpy()
            a = np.where(cur.reshape((1, -1) + (1,) * (a.ndim - 2)), a, b)
        if name == "DE" and self.de_epl:
            lane, slot = np.arange(self.d) % 4, np.arange(self.d) // 4
            pos = 8 * (slot // 2) + 2 * lane + slot % 2                # where parameter i sits in a row (ptmi_de_row_stride)
            return np.ascontiguousarray(a[..., pos])
        if name == "AM" and self.am_pos is not None:
            return np.ascontiguousarray(a[..., self.am_pos])              # parameter order whatever the device format
        return a.view(np.uint64) if name in ("nacc", "jstat", "nswap", "AMflag") else a

    def put(self, name, value):
        torch = _torch()
        if name == "AM":
            value = self.am_rows(np.asarray(value))                      # parameter order in, the buffer's row format on the device
            if self.am_rle:
                self.t["AMflag"].fill_(_lib.AMROW_KEY)                    # rows written from outside are KEY rows
        self.t[name].copy_(torch.from_numpy(np.ascontiguousarray(value)).to(self.t[name].dtype))

    def by_temp(self, name):
        """A per-slot array ([W][nt][...]) reordered by temperature rank."""
        a, so = self.get(name), self.get("slot_of").astype(np.int64)
        return np.take_along_axis(a, so.reshape(so.shape + (1,) * (a.ndim - 2)), 1)

    def sync(self):
        _lib.check(self.lib.ptmi_sync(self.h))
        if self.stats_async:
            self._st.synchronize()
        self._check_sytrd_info()

    def _check_sytrd_info(self):
        """eig_mode "sytrd": the divide-and-conquer solver's convergence word of the last factorization that has finished (it follows
        the result to pinned host memory on the factorization's stream: reading it never waits)."""
        if self.eig_mode == "sytrd":
            v = C.c_int32(0)
            _lib.check(self.lib.ptmi_eig_sytrd_info(self.h, C.byref(v)))
            if v.value != 0:
                raise _lib.PtmiError("ptmi_eig_sytrd: the tridiagonal eigensolver did not converge (info = %d); the eigenvectors of "
                                     "that covariance epoch are not valid" % v.value)

    # ------------------------------------------------------------------ set-up
    def _eig_host(self, w, cov):
        """U, S of the jump covariance by LAPACK (see factorize)."""
        for gi, g in enumerate(self.groups):                          # per group, :139-145 and :797-803
            U, S = factorize(cov if self.whole else cov[np.ix_(g, g)], self.per_walker)
            self.put_eig(U, S, w, gi)

    def _eig_host_pooled(self):
        """The pooled covariance's factorization on the host (factorize(), the same bits as _eig_host) with the GPU idle for as
        short as possible: the matrix comes down into pinned memory, the eigenvector rows and eigenvalues go up from pinned
        memory, one wait in all (the download's); the uploads are queued and the next launch behind them."""
        self._eig_begin()
        self._eig_end()

    def _eig_begin(self, stream=None):
        """First half of _eig_host_pooled: the covariance sets out for pinned host memory behind the statistics kernels (on `stream`:
        the engine's, or the side stream the statistics run on)."""
        torch = _torch()
        d = self.d
        stream = self.stream if stream is None else stream
        if getattr(self, "_pin", None) is None:
            self._pin = (torch.empty((d, d), dtype=torch.float64).pin_memory(), torch.empty((d, d), dtype=torch.float64).pin_memory(),
                         torch.empty(d, dtype=torch.float64).pin_memory(), torch.cuda.Event())
        cov_h, ut_h, s_h, ev = self._pin
        with torch.cuda.stream(stream):
            cov_h.copy_(self.t["cov"][0], non_blocking=True)
            ev.record(stream)
        self._eig_pending = True

    def _eig_end(self):
        """Second half: wait for the covariance (NOT for the stream: with eig_lag = 1 a launch is running meanwhile), factorize,
        queue the uploads; whatever is launched next reads the new table."""
        torch = _torch()
        cov_h, ut_h, s_h, ev = self._pin
        with torch.cuda.stream(self.stream):
            ev.synchronize()
            U, S = factorize(cov_h.numpy(), False)
            np.copyto(ut_h.numpy(), U.T)
            np.copyto(s_h.numpy(), S)
            self.t["Ut"][0, 0].copy_(ut_h, non_blocking=True)
            self.t["S"][0, 0].copy_(s_h, non_blocking=True)
        self._eig_pending = False
        self.eig_epochs += 1

    def _eig_begin_side(self):
        """eig_mode "hipsolver" with eig_lag > 0: the library's eigensolver on a SIDE stream, behind the statistics kernels and
        beside the launches that follow (its thousands of small kernels fill what the step kernel leaves free); the result waits in
        staging tensors until _eig_end_side."""
        torch = _torch()
        if getattr(self, "_side", None) is None:
            # stats_async: the factorization follows the statistics on THEIR stream (which waits for the period's last launch itself)
            self._side = self._st if self.stats_async else torch.cuda.Stream(device=self.device, priority=-1)      # its small kernels go ahead of the step kernel's next blocks (priority 0: 7.1e8 against 7.3e8 at config 4)
            self._side_go, self._side_done = torch.cuda.Event(), torch.cuda.Event()
            # the staging tensors start as the table in force: whatever goes wrong on the side, they never hold garbage
            self._ut_next, self._s_next = self.t["Ut"].clone(), self.t["S"].clone()
            self._cov_side = torch.empty_like(self.t["cov"])          # the covariance the factorization reads: the next epoch's statistics may overwrite t["cov"] meanwhile
        with torch.cuda.stream(self._st if self.stats_async else self.stream):
            self._cov_side.copy_(self.t["cov"])                       # behind the statistics, on their stream
        if not self.stats_async:
            self._side_go.record(self.stream)
            self._side.wait_event(self._side_go)
        self._side_err = None

        def work():
            # on a thread of its own: the library's driver waits on the host for its status word, which would keep this
            # thread from queueing the launches the factorization is meant to run beside.  The thread reports through
            # self._side_err (the library's last-error text is per thread: it is read HERE, not by the thread that joins)
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self._side):
                    if self.eig_mode == "sytrd":
                        _lib.check(self.lib.ptmi_eig_sytrd_from(self.h, C.c_void_p(self._side.cuda_stream), C.c_void_p(self._cov_side.data_ptr()),
                                                                C.c_void_p(self._ut_next.data_ptr()), C.c_void_p(self._s_next.data_ptr())))
                    elif self.eig_mode == "ql":               # every walker's matrix (or the pooled one): chains of rotations, little of the GPU each
                        _lib.check(self.lib.ptmi_eig_ql_from(self.h, C.c_void_p(self._side.cuda_stream), C.c_void_p(self._cov_side.data_ptr()),
                                                             C.c_void_p(self._ut_next.data_ptr()), C.c_void_p(self._s_next.data_ptr())))
                    else:
                        w, V = torch.linalg.eigh(self._cov_side)
                        w, order = w.abs().sort(dim=-1, descending=True, stable=True)
                        self._ut_next[:, 0].copy_(torch.gather(V, -1, order.unsqueeze(-2).expand_as(V)).transpose(-1, -2))
                        self._s_next[:, 0].copy_(w)
            except BaseException as e:              # noqa: B902 -- re-raised by the thread that joins (_eig_end_side)
                self._side_err = e
            finally:
                self._side_done.record(self._side)

        import threading
        self._side_thread = threading.Thread(target=work, daemon=True)
        self._side_thread.start()
        self._eig_pending = True

    def _eig_end_side(self):
        torch = _torch()
        self._side_thread.join()
        self._eig_pending = False
        if self._side_err is not None:                # the table in force stays; the run does not go on with a half-made one
            err, self._side_err = self._side_err, None
            raise _lib.PtmiError("the factorization on the side stream failed: %r" % (err,)) from err
        self.stream.wait_event(self._side_done)
        with torch.cuda.stream(self.stream):
            self.t["Ut"].copy_(self._ut_next)
            self.t["S"].copy_(self._s_next)
        self.eig_epochs += 1

    def _eig_finish(self):
        """The pending factorization of the last covariance epoch takes effect (eig_lag launches after it, or at the next epoch)."""
        if self._eig_pending:
            if self.eig_mode in ("hipsolver", "sytrd", "ql"):
                self._eig_end_side()
            else:
                self._eig_end()
        self._eig_wait = 0

    def _eig_hipsolver(self):
        """U, S of every covariance the engine holds by the ROCm library's symmetric eigensolver, on the stream (factorize()'s
        pooled rule: eigenvalues by decreasing size and in absolute value, eigenvectors as the rows of Ut)."""
        torch = _torch()
        with torch.cuda.stream(self.stream):
            if not self.whole:
                # parameter groups (PTMCMCSampler.py:139-145, 797-803: one factorization per group's block of the covariance), any ndim:
                # the block gathered on the device, the vectors embedded in the full space, one per row of the group's table
                cov = self.t["cov"]
                self.t["Ut"].zero_()
                self.t["S"].zero_()
                for gi, g in enumerate(self.groups):
                    idx = torch.as_tensor(g, device=self.device)
                    m = len(g)
                    w, V = torch.linalg.eigh(cov[:, idx][:, :, idx])                           # [Wc][m], [Wc][m][m]
                    w, order = w.abs().sort(dim=-1, descending=True, stable=True)
                    Vt = torch.gather(V, -1, order.unsqueeze(-2).expand_as(V)).transpose(-1, -2)    # rows = eigenvectors
                    rows = self.t["Ut"][:, gi, :m]                                             # [Wc][m][d] (a view)
                    rows[:, :, idx] = Vt
                    self.t["S"][:, gi, :m] = w
                return
            w, V = torch.linalg.eigh(self.t["cov"])                  # [Wc][d], [Wc][d][d] (columns)
            # by decreasing |eigenvalue| (a covariance's are >= 0 up to rounding: a slightly negative one must not jump the queue)
            w, order = w.abs().sort(dim=-1, descending=True, stable=True)
            self.t["Ut"][:, 0].copy_(torch.gather(V, -1, order.unsqueeze(-2).expand_as(V)).transpose(-1, -2))
            self.t["S"][:, 0].copy_(w)

    def _eig_host_all(self, cov):
        """Per-walker mode: the W independent factorizations, one upload for all.  From 64 walkers on they run on a
        pool of spawned host processes (numpy only; every LAPACK call single-threaded, so each walker's bits are those
        of a lone reference run); PTMI_EIG_WORKERS sets the pool size (0 = in this process)."""
        import os
        from . import _eigworker
        torch = _torch()
        groups = None if self.whole else self.groups
        nwork = int(os.environ.get("PTMI_EIG_WORKERS", min(64, _usable_cores())))
        if self.Wc < 64 or nwork <= 1:
            Ut, Sv = _eigworker.svd_chunk((cov, groups))
        else:
            if getattr(self, "_eig_pool", None) is None:
                import multiprocessing as mp
                self._eig_pool = mp.get_context("spawn").Pool(nwork)       # spawn: no fork of a process that holds HIP state
            bounds = np.linspace(0, self.Wc, min(self.Wc, 4 * nwork) + 1).astype(int)
            parts = self._eig_pool.map(_eigworker.svd_chunk, [(cov[a:b], groups) for a, b in zip(bounds[:-1], bounds[1:]) if b > a])
            Ut, Sv = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        self.t["Ut"].copy_(torch.from_numpy(Ut))
        self.t["S"].copy_(torch.from_numpy(Sv))

    def put_eig(self, U, S, w=0, gi=0):
        """Upload a group's eigenvectors (columns of U) embedded in the full space, one per row."""
        torch = _torch()
        g = self.groups[gi]
        Ut = np.zeros((self.d, self.d))
        Sv = np.zeros(self.d)
        Ut[np.ix_(np.arange(len(g)), g)] = np.asarray(U).T
        Sv[:len(g)] = S
        self.t["Ut"][w, gi].copy_(torch.from_numpy(Ut))
        self.t["S"][w, gi].copy_(torch.from_numpy(Sv))

    def init_state(self, p0, i0=0):
        """Point(s) the chains hold at iteration ``i0``: ``p0`` of shape [d] (broadcast) or [W][nt][d] (by slot)."""
        torch = _torch()
        p0 = np.asarray(p0, dtype=np.float64)
        full = np.array(p0 if p0.ndim == 3 else np.broadcast_to(p0, (self.W, self.nt, self.d)))
        self.t["X"].copy_(torch.from_numpy(full))
        if self.t.get("sloc") is not None:
            self.t["sloc"].zero_()                                    # every state is in X (a segment of the split path abandoned half way left it otherwise)
        _lib.check(self.lib.ptmi_eval_state(self.h))                 # :479-487
        self._store_initial(i0)
        self.iter = int(i0)

    def am_rows(self, rows):
        """Rows in parameter order (last axis) -> the AM buffer's row format (a torch tensor or a numpy array)."""
        if self.am_inv is None:
            return rows
        if isinstance(rows, np.ndarray):
            return rows[..., self.am_inv]
        return rows[..., _torch().from_numpy(self.am_inv).to(rows.device)]

    def am_params(self, rows):
        """Rows of the AM buffer (last axis in its row format) -> parameter order."""
        if self.am_pos is None:
            return rows
        if isinstance(rows, np.ndarray):
            return rows[..., self.am_pos]
        return rows[..., _torch().from_numpy(self.am_pos).to(rows.device)]

    def am_expand(self, w0=0, nw=None, it_lo=None, it_hi=None):
        """Every row of the AM ring, in place (``ptmi_am_expand``): the rows of rejected steps, which ``am_mode="rle"`` does not
        store, are copied forward from the row before them, for iterations ``it_lo .. it_hi`` of walkers ``w0 .. w0 + nw - 1``;
        default: the current covariance period up to the current iteration (``am_period``) -- once the ring wraps, the repeats of
        an older period have lost the row they hang on.  A no-op with ``am_mode="rows"``."""
        if not self.am_rle:
            return
        it_hi = self.iter if it_hi is None else int(it_hi)
        it_lo = self.am_period(it_hi)[0] if it_lo is None else int(it_lo)
        _lib.check(self.lib.ptmi_am_expand(self.h, int(w0), self.W - int(w0) if nw is None else int(nw), it_lo, it_hi))

    def am_period(self, it=None):
        """(first, last) iteration of the covariance period the ring holds at iteration ``it``: [E, it], E = the last multiple of
        covUpdate below ``it`` (row 0 of the ring, until iteration E + covUpdate overwrites it)."""
        it = self.iter if it is None else int(it)
        E = ((it - 1) // self.cov_update) * self.cov_update if it > 0 else 0
        return E + (1 if it - E >= self.cov_update else 0), it

    def _store_initial(self, i0=0):
        """updateChains(p0, lnlike0, lnprob0, i0), :491: row i0 % covUpdate of the AM ring holds the point."""
        torch = _torch()
        if self.owns_cold:
            ar = torch.arange(self.W, device=self.device)
            idx = self.t["slot_of"][:, 0].long()
            row = int(i0) % self.cov_update
            self.t["AM"][:, row, :] = self.am_rows(self.t["X"][ar, idx])
            if self.am_rle:
                self.t["AMflag"][:, row] = _lib.AMROW_KEY
            if self.t["AMaux"] is not None:
                self.t["AMaux"][:, row, 0] = self.t["lnL"][ar, idx]
                self.t["AMaux"][:, row, 1] = self.t["lp"][ar, idx]

    # ------------------------------------------------------------------ epochs
    def update_cov(self, it_done):
        """Covariance epoch after iteration ``it_done`` (:545-560): device Welford, host SVD."""
        if not self.owns_cold:
            return
        # A factorization still pending from the epoch before (eig_lag >= the launches of a period) is finished here at the latest.
        # On the side stream (device factorizations) it goes on BESIDE this epoch's statistics, which do not read the table: they are
        # queued first, the wait for the old table comes behind them -- the same table in force for the same launches either way.
        late = self._eig_pending and self.late_finish
        if not late:
            self._eig_finish()
        self._check_sytrd_info()                                      # how the last finished factorization went
        if self.stats_async:
            self._update_cov_async(it_done)
            return
        _lib.check(self.lib.ptmi_update_cov(self.h, it_done))
        if late:
            self._eig_finish()
        if self.am_rle and self.t["DE"] is not None and self.burn % self.cov_update != 0:
            # a DE epoch (every `burn` iterations, :563-571) copies ALL covUpdate ring rows; unless burn is a multiple of covUpdate
            # some of them belong to the period that ends here, and am_expand only reaches back to the current period's start:
            # copy that period's repeats forward now, before the ring wraps (the statistics above have taken the run lengths)
            self.am_expand(it_lo=self.am_period(it_done)[0], it_hi=it_done)
        if self.eig_mode == "ql" and self.eig_lag:
            self._eig_begin_side()                                    # beside the launches that follow; run() puts it into force
            self._eig_wait = self.eig_lag
            return
        if self.eig_mode in ("jacobi", "ql"):                         # stays on the stream: no host synchronisation
            _lib.check(self.lib.ptmi_eig_jacobi(self.h) if self.eig_mode == "jacobi" else self.lib.ptmi_eig_ql(self.h))
            self.eig_epochs += 1
            return
        if self.eig_mode in ("hipsolver", "sytrd"):
            if self.eig_lag and not self.per_walker:
                self._eig_begin_side()                                # run() puts it into force eig_lag launches later
                self._eig_wait = self.eig_lag
                return
            if self.eig_mode == "sytrd":
                _lib.check(self.lib.ptmi_eig_sytrd(self.h, None, None, None))      # on the engine's stream, into Ut / S
            else:
                self._eig_hipsolver()
            self.eig_epochs += 1
            return
        if self.Wc == 1 and not self.per_walker and self.whole:
            if self.eig_lag:
                self._eig_begin()                                     # run() finishes it behind the eig_lag-th launch from here
                self._eig_wait = self.eig_lag
            else:
                self._eig_host_pooled()
            return
        elif self.Wc == 1:
            self._eig_host(0, self.get("cov")[0])
        else:
            self._eig_host_all(self.get("cov"))
        self.eig_epochs += 1

    def _update_cov_async(self, it_done):
        """The covariance epoch with the statistics BESIDE the launches that follow (stats_async): the period that ends here stays in
        its ring, the launches from here on write the other one, the side stream takes the statistics and the factorization."""
        torch = _torch()
        old = {k: self.t[k] for k in ("AM", "AMaux", "AMflag")}
        with torch.cuda.stream(self.stream):
            if self._stats_queued:
                self.stream.wait_event(self._ev_stats)                # the other ring is free once the statistics that read it are done (a period ago)
            # ring row 0 holds the period's last iteration: every reader of the next period starts from it (am_expand's walk back,
            # am_period); it moves along with its flag
            for k, v in old.items():
                if v is not None:
                    self._alt[k][:, 0] = v[:, 0]
            self._ev_period.record(self.stream)
        for k, v in old.items():
            self.t[k], self._alt[k] = self._alt[k], v
        ptr = lambda v: C.c_void_p(v.data_ptr()) if v is not None else None      # noqa: E731
        _lib.check(self.lib.ptmi_set_am_buffers(self.h, ptr(self.t["AM"]), ptr(self.t["AMaux"]), ptr(self.t["AMflag"])))
        self._st.wait_event(self._ev_period)
        _lib.check(self.lib.ptmi_update_cov_on(self.h, it_done, C.c_void_p(self._st.cuda_stream), ptr(old["AM"]), ptr(old["AMflag"])))
        self._ev_stats.record(self._st)
        self._stats_queued = True
        if self.eig_mode in ("hipsolver", "sytrd"):
            self._eig_begin_side()
        else:
            self._eig_begin(self._st)
        self._eig_wait = self.eig_lag

    def update_de(self, it_done=None):
        if self.owns_cold and self.t["DE"] is not None:
            if it_done is not None:
                self.am_expand(it_hi=it_done)                          # the DE history takes every row of the ring (a no-op with stored rows)
            _lib.check(self.lib.ptmi_update_de(self.h))
            self.de_head = (self.de_head + min(self.cov_update, self.burn)) % self.burn

    def set_de_active(self, on=True):
        _lib.check(self.lib.ptmi_set_de_active(self.h, int(on)))
        self.de_on = bool(on)

    def set_de_head(self, head):
        _lib.check(self.lib.ptmi_set_de_head(self.h, int(head)))
        self.de_head = int(head)

    # ------------------------------------------------------------------ checkpoint
    def checkpoint(self):
        """Everything needed to continue bit-identically: device arrays, ring head, DE flag, iteration.
        (The RNG is counter based: its whole state is the iteration number.)"""
        self.sync()
        st = {"t_" + k: v.cpu().numpy() for k, v in self.t.items() if v is not None and k not in ("Q", "qaux", "Q2", "sloc")}
        st.update(iter=self.iter, de_on=int(self.de_on), de_head=self.de_head, swap_proposed=self.swap_proposed,
                  eig_epochs=self.eig_epochs,
                  # eig_lag: an epoch's table that is not in force yet (its covariance is in t_cov; restore() factorizes it again)
                  eig_pending=int(self._eig_pending), eig_wait=int(self._eig_wait))
        if self.stats_async:
            st.update({"alt_" + k: v.cpu().numpy() for k, v in self._alt.items() if v is not None})
        return st

    def restore(self, st):
        torch = _torch()
        # whatever this engine has under way is void -- and is brought to an end FIRST: a factorization on the side stream (its thread
        # joined, its result dropped), statistics still running beside the launches (stats_async: they write cov / mu / M2)
        if self._eig_pending:
            try:
                self._eig_finish()
            except _lib.PtmiError:
                pass
        if self.stats_async:
            self._st.synchronize()
            self._stats_queued = False
        for k, v in self.t.items():
            if v is not None and "t_" + k in st:
                v.copy_(torch.from_numpy(np.ascontiguousarray(st["t_" + k])).to(v.dtype))
        if self.stats_async:                                          # the ring of the period before (readers of older rows: get("AM"))
            for k, v in self._alt.items():
                if v is not None and "alt_" + k in st:
                    v.copy_(torch.from_numpy(np.ascontiguousarray(st["alt_" + k])).to(v.dtype))
        self.iter, self.swap_proposed = int(st["iter"]), int(st["swap_proposed"])
        self.eig_epochs = int(st["eig_epochs"])                       # (behind _eig_finish, which counts the dropped table)
        if self.t["DE"] is not None:
            self.set_de_head(int(st["de_head"]))
            if int(st["de_on"]):
                self.set_de_active(True)
        if int(st.get("eig_pending", 0)):
            # the checkpoint fell between a covariance epoch and the launch its table takes effect at: the factorization is
            # issued again from the saved covariance and becomes effective after the same number of launches
            if self.eig_mode in ("hipsolver", "sytrd", "ql"):
                self._eig_begin_side()
            else:
                self._eig_begin()
            self._eig_wait = int(st["eig_wait"])

    # pieces of the swap for a ladder sharded over GPUs (see sharded.py)
    def gather_lnl(self, out):
        _lib.check(self.lib.ptmi_swap_gather_lnl(self.h, out.data_ptr()))

    def sweep(self, it, lnl_glob, map_out):
        _lib.check(self.lib.ptmi_swap_sweep(self.h, it, lnl_glob.data_ptr(), map_out.data_ptr()))

    def write_am(self, it):
        _lib.check(self.lib.ptmi_swap_write_am(self.h, it))

    # device-side exchange (no host synchronisation), see include/ptmi.h
    def sweep_blocks(self, it, lnl_blocks, map_out):
        _lib.check(self.lib.ptmi_swap_sweep_blocks(self.h, it, lnl_blocks.data_ptr(), map_out.data_ptr()))

    def exchange_pack(self, map_dev, send):
        _lib.check(self.lib.ptmi_exchange_pack(self.h, map_dev.data_ptr(), send.data_ptr()))

    def exchange_apply(self, recv):
        _lib.check(self.lib.ptmi_exchange_apply(self.h, recv.data_ptr()))

    def exchange_multihop(self):
        """After exchange_pack: does any row of this sweep travel beyond a neighbouring block? (same answer on every GPU)"""
        v = C.c_int32(0)
        _lib.check(self.lib.ptmi_exchange_multihop(self.h, C.byref(v)))
        return bool(v.value)

    def exchange_violations(self):
        v = C.c_int32(0)
        _lib.check(self.lib.ptmi_exchange_status(self.h, C.byref(v)))
        return v.value

    def _epochs(self, it):
        cu, burn = self.cov_update, self.burn
        de_now = (it - 1) % burn == 0 and it - 1 != 0
        if de_now and self.stats_async:
            self.update_de(it - 1)                                    # reads the ring of the period that ends here: before the switch
        if (it - 1) % cu == 0 and it - 1 != 0:
            self.update_cov(it - 1)
        if de_now and not self.stats_async:
            self.update_de(it - 1)                                    # :563-571
        if it - 1 == burn and self.weights[2] > 0 and self.t["DE"] is not None:
            self.set_de_active(True)                                  # :574-585

    def _segment_end(self, it, last):
        """Last iteration of the launch that starts at ``it``: no epoch inside, swap only at its end."""
        end = last
        pers = [self.cov_update, self.burn]
        if self.tskip > 0 and self.ntg > 1:
            pers.append(self.tskip)
        for per in pers:
            end = min(end, ((it - 1) // per + 1) * per)
        return end

    # ------------------------------------------------------------------ stepping
    def mh_steps(self, iter0, nsteps):
        _lib.check(self.lib.ptmi_mh_steps(self.h, iter0, nsteps))

    def last_variant(self):
        """Flags of the fused-kernel instantiation the last ``mh_steps`` launched (``_lib.VAR_*``), lanes, slots."""
        v = C.c_int32(0)
        _lib.check(self.lib.ptmi_last_mh_variant(self.h, C.byref(v)))
        return (v.value & 0xFFF) | (v.value & _lib.VAR_UTPAD), (v.value >> 12) & 0xFF, (v.value >> 20) & 0xFF

    def swap(self, it):
        """PT swap of iteration ``it`` with the whole ladder on this GPU (:631-697)."""
        _lib.check(self.lib.ptmi_swap(self.h, it))
        self.swap_proposed += 1

    def run(self, niter):
        """Advance ``niter`` iterations (:499-503 for every chain)."""
        last = self.iter + niter
        it = self.iter + 1
        while it <= last:
            self._epochs(it)
            end = self._segment_end(it, last)
            self.mh_steps(it, end - it + 1)
            if self.tskip > 0 and self.ntg > 1 and end % self.tskip == 0:
                self.swap(end)
            if self._eig_pending:                                     # eig_lag launches after the epoch its table takes effect
                self._eig_wait -= 1
                # (when the next iteration opens a covariance epoch, update_cov finishes it: behind that epoch's statistics)
                if self._eig_wait <= 0 and not (end % self.cov_update == 0 and end + 1 <= last and self.owns_cold):
                    self._eig_finish()
            it = end + 1
        self.iter = last

    # ------------------------------------------------------------------ batched callbacks
    def _cb_values(self, v):
        """A callback's return value as a contiguous f64 tensor of W * nt values on this GPU (no copy when it already is one)."""
        torch = _torch()
        if not (torch.is_tensor(v) and v.dtype == torch.float64 and v.device == self.device and v.is_contiguous() and v.numel() == self.W * self.nt):
            v = torch.as_tensor(v, dtype=torch.float64, device=self.device).reshape(self.W, self.nt).contiguous()
        return v

    def eval_callback(self, X, logl, logp):
        """logp then logl of every row of the device tensor X [W][nt][d] through BATCHED callbacks
        ``f(X[n, d]) -> [n]`` (torch tensors on this GPU in, the same out): the device-side form of the reference's
        ``_function_wrapper`` boundary (PTMCMCSampler.py:1072-1086, called at :605-611).  Nothing is copied to the host.
        ``logp=None`` is the flat prior (no launch at all).  Where the prior is -inf the likelihood value is not used: the reference
        does not even call it (:607-608), here the accept test never reads it (ptmi_accept: -inf prior => -inf posterior)."""
        flat = X.view(-1, self.d)
        if logp is None:
            if getattr(self, "_lp_zero", None) is None:
                self._lp_zero = _torch().zeros((self.W, self.nt), dtype=_torch().float64, device=self.device)
            lp = self._lp_zero
        else:
            lp = self._cb_values(logp(flat))
        return self._cb_values(logl(flat)), lp

    def init_state_callback(self, p0, logl, logp, i0=0):
        """init_state for a likelihood that lives in a batched callback (:479-487)."""
        torch = _torch()
        p0 = np.asarray(p0, dtype=np.float64)
        full = np.array(p0 if p0.ndim == 3 else np.broadcast_to(p0, (self.W, self.nt, self.d)))
        self.t["X"].copy_(torch.from_numpy(full))
        if self.t.get("sloc") is not None:
            self.t["sloc"].zero_()                                    # every state is in X
        ll, lp = self.eval_callback(self.t["X"], logl, logp)
        ll, lp = ll.reshape(self.W, self.nt), lp.reshape(self.W, self.nt)
        self.t["lnL"].copy_(torch.where(torch.isneginf(lp), lp, ll))               # :481-483
        self.t["lp"].copy_(lp)
        self._store_initial(i0)
        self.iter = int(i0)

    def split_step(self, it, logl, logp):
        """One iteration of every chain with batched callbacks: ptmi_propose -> callbacks on the device tensor of
        proposals -> ptmi_accept.  All on the engine's stream; no host copy of the proposals."""
        if self.t["Q"] is None:
            raise _lib.PtmiError("the callback path needs the engine built with split=True")
        _lib.check(self.lib.ptmi_propose(self.h, it))
        ll, lp = self.eval_callback(self.t["Q"], logl, logp)
        _lib.check(self.lib.ptmi_accept(self.h, it, ll.data_ptr(), lp.data_ptr()))

    def builtin_logl(self):
        """The built-in isotropic Gaussian as a batched CALLBACK ``f(X[n, d]) -> [n]`` (``ptmi_rows_logl``: a device kernel behind the
        C ABI, the fused kernels' bits): with it ``run_callback`` reproduces ``run`` bit for bit."""
        torch = _torch()

        def logl(X):
            out = torch.empty(X.shape[0], dtype=torch.float64, device=X.device)
            _lib.check(self.lib.ptmi_rows_logl(self.h, X.data_ptr(), X.shape[0], out.data_ptr()))
            return out

        return logl

    def dense_logl_callback(self, mu, P):
        """The dense Gaussian -(x - mu)^T P (x - mu) / 2 (the reference's own test likelihood, tests/test_simple.py:14-41) as a batched
        CALLBACK whose product runs on the matrix cores for the whole batch at once: (X - mu) @ P is ONE f64 GEMM of [n, d] x [d, d]
        (hipBLASLt through torch), then a row-wise dot.  The built-in ``("dense", mu, P)`` family keeps its table in LDS up to 104
        parameters; beyond, its fused kernels stream the d x d table per chain-step (6.5 s per 100 steps at 1000-d) -- this callback
        on the split path is the fast way there.  Its sums have the library's order, not the oracle's: a callback
        like a user's own, checked against NumPy."""
        torch = _torch()
        mu_t = torch.as_tensor(np.asarray(mu, dtype=np.float64), device=self.device)
        P_t = torch.as_tensor(np.ascontiguousarray(np.asarray(P, dtype=np.float64)), device=self.device)

        def logl(X):
            R = X - mu_t
            return torch.mm(R, P_t).mul_(R).sum(-1).mul_(-0.5)

        return logl

    def proposals(self):
        """The device tensor that holds the current proposals (``ptmi_proposals``): Q, or Q2 in turn after ``ptmi_accept_propose``."""
        p = C.c_void_p()
        _lib.check(self.lib.ptmi_proposals(self.h, C.byref(p)))
        q2 = self.t["Q2"]
        return q2 if q2 is not None and p.value == q2.data_ptr() else self.t["Q"]

    def callback_segment(self, it, end, logl, logp):
        """Iterations ``it .. end`` with nothing between them (no epoch, no swap: a segment of ``run``): ONE proposal launch, then per
        iteration the callbacks on the device tensor of proposals and ptmi_accept_propose -- the accept test of iteration j and the
        proposal of j + 1 in one launch, a chain's row in once and out once (csrc/ptmi_split.hip) -- and ptmi_accept behind the last.
        The same chains as ``split_step`` iteration by iteration, bit for bit."""
        if self.t["Q"] is None:
            raise _lib.PtmiError("the callback path needs the engine built with split=True")
        lib, h = self.lib, self.h
        # cycles with AM entries: the picks' increments for a piece of the segment at a time, on the matrix cores ahead of the proposals
        if getattr(self, "_am_piece", None) is None:
            v = C.c_int32(0)
            _lib.check(lib.ptmi_split_am_piece(h, C.byref(v)))
            self._am_piece = v.value
        piece = self._am_piece
        if piece:
            _lib.check(lib.ptmi_split_am_prepare(h, it, min(piece, end - it + 1)))
        _lib.check(lib.ptmi_propose(h, it))
        for j in range(it, end):
            ll, lp = self.eval_callback(self.proposals(), logl, logp)
            if piece and (j + 1 - it) % piece == 0:                   # the proposal of j + 1 opens the next piece
                _lib.check(lib.ptmi_split_am_prepare(h, j + 1, min(piece, end - j)))
            _lib.check(lib.ptmi_accept_propose(h, j, ll.data_ptr(), lp.data_ptr()))       # (between here and ptmi_accept X is not the state: sloc)
        ll, lp = self.eval_callback(self.proposals(), logl, logp)
        _lib.check(lib.ptmi_accept(h, end, ll.data_ptr(), lp.data_ptr()))

    def callback_segment_graph(self, it, end, logl, logp):
        """``callback_segment`` as ONE hipGraph launch.  At small batches the segment is launch-bound -- two launches of a few
        microseconds per iteration, each behind a Python call -- so its launches are captured once per segment LENGTH (torch's graph
        capture on a side stream the library's stream is pointed at meanwhile; the iteration comes from a counter in device memory:
        ``ptmi_device_iter``) and replayed for every later segment of that length: set the counter, launch the graph.  The callbacks
        must be graph-safe (no host synchronisation, the same launches for every batch: any fixed torch expression or device
        kernel is).  What the captured launches bake in -- the DE ring's head, whether DE is in the cycle -- is part of the cache
        key.  Returns False where it does not apply (AM entries in the cycle: their increments are listed on the host's iteration;
        configurations the row kernels do not serve): the caller then runs ``callback_segment``.  Same results, bit for bit."""
        torch = _torch()
        if self.t["Q2"] is None or self.weights[1] > 0:
            return False
        if getattr(self, "_graphs", None) is None:
            self._graphs = {}
        L = end - it + 1
        key = (L, bool(self.de_on), int(self.de_head), id(logl), id(logp))
        g = self._graphs.get(key)
        lib, h = self.lib, self.h
        if g is None:
            if len(self._graphs) >= 64:               # (callbacks that are new objects at every call would capture for ever: keep the SAME callables)
                self._graphs.clear()
            # a warm-up pass of the callbacks outside the capture (lazy initialisations of the libraries behind them), then the capture
            self.eval_callback(self.t["Q"], logl, logp)
            torch.cuda.synchronize(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(self.stream)
            g = torch.cuda.CUDAGraph()
            _lib.check(lib.ptmi_device_iter(h, 1))
            _lib.check(lib.ptmi_set_stream(h, C.c_void_p(side.cuda_stream)))
            try:
                with torch.cuda.graph(g, stream=side):
                    _lib.check(lib.ptmi_propose(h, 0))
                    for j in range(L - 1):
                        ll, lp = self.eval_callback(self.proposals(), logl, logp)
                        _lib.check(lib.ptmi_accept_propose(h, j, ll.data_ptr(), lp.data_ptr()))
                    ll, lp = self.eval_callback(self.proposals(), logl, logp)
                    _lib.check(lib.ptmi_accept(h, L - 1, ll.data_ptr(), lp.data_ptr()))
            finally:
                _lib.check(lib.ptmi_set_stream(h, C.c_void_p(self.stream.cuda_stream)))
                _lib.check(lib.ptmi_device_iter(h, 0))
            g = (g, 1 if self.proposals() is self.t["Q2"] else 0)    # ... and which buffer holds the segment's last proposals
            self._graphs[key] = g
        _lib.check(lib.ptmi_set_device_iter(h, it))
        g[0].replay()
        _lib.check(lib.ptmi_set_proposals(h, g[1]))
        return True

    def run_callback(self, niter, logl, logp, fused=True, graph=False):
        """``run`` with the likelihood and the prior in batched callbacks: the same segments (epochs and swaps between them, the
        late table of ``eig_lag`` counted in segments as ``run`` counts it in launches).  ``fused=False``: propose / accept as two
        launches per iteration (``split_step``; same results).  ``graph=True``: every segment one hipGraph launch
        (``callback_segment_graph``: for small, launch-bound batches with graph-safe callbacks; same results)."""
        last = self.iter + niter
        it = self.iter + 1
        while it <= last:
            self._epochs(it)
            end = self._segment_end(it, last)
            if fused and graph and self.callback_segment_graph(it, end, logl, logp):
                pass
            elif fused:
                self.callback_segment(it, end, logl, logp)
            else:
                for j in range(it, end + 1):
                    self.split_step(j, logl, logp)
            if self.tskip > 0 and self.ntg > 1 and end % self.tskip == 0:
                self.swap(end)
            if self._eig_pending:
                self._eig_wait -= 1
                if self._eig_wait <= 0 and not (end % self.cov_update == 0 and end + 1 <= last and self.owns_cold):
                    self._eig_finish()
            it = end + 1
        self.iter = last

    # ------------------------------------------------------------------ timing
    def timer_start(self):
        _lib.check(self.lib.ptmi_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_double(0)
        _lib.check(self.lib.ptmi_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def lnprob(self):
        """beta * lnL + lp of every slot (the reference's lnprob0)."""
        beta = (1 / self.temps_mh)[self.get("temp_of")]
        return beta * self.get("lnL") + self.get("lp")
