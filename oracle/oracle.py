"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
import this module.  It wraps ``libptmcmc_oracle.so`` (built from
``ptmcmc_oracle.c`` by ``make -C oracle``) and composes the per-operation C
functions into the sampler loop of the reference
(``PTMCMCSampler.py:495-528`` driving ``PTMCMCOneStep`` ``:530-629``) for all
temperatures of ``nwalkers`` independent replicas in one process.

Draw sources: ``replay`` (the reference's recorded draws, one stream per rank;
pins the oracle to the reference) or Philox (what the HIP kernels use).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libptmcmc_oracle.so")

LOGL = {"iso": 0, "dense": 1, "curved": 2, "interval": 3}
LOGP = {"flat": 0, "box": 1}
J_SCAM, J_AM, J_DE, J_NUTS, J_HMC, J_NTYPES = 0, 1, 2, 3, 4, 5
J = {"scam": 0, "am": 1, "de": 2, "nuts": 3, "hmc": 4}
K_INT, K_UNI, K_NRM, K_SHUF, K_EXP = 0, 1, 2, 3, 4
GJ_EPS, GJ_MU, GJ_HBAR, GJ_EPSBAR, GJ_NITER, GJ_HITER, GJ_HAVE_EPS, GJ_NSTATE = 0, 1, 2, 3, 4, 5, 6, 8

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint64)


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "ndim", "ntemps", "nwalkers", "lanes", "logl_kind", "logp_kind", "w_scam", "w_am", "w_de",
        "de_on", "de_size", "cov_update", "tskip", "cov_per_walker", "ntemps_global", "temp0", "walker0", "ngroups")] + [
        ("seed", C.c_uint64), ("logl_par", _dp), ("logp_par", _dp), ("temps_mh", _dp), ("beta", _dp),
        ("gsize", _ip), ("gmask", _dp)] + [(n, C.c_int32) for n in (
        "w_nuts", "w_hmc", "gj_nburn", "hmc_min", "hmc_max", "nuts_maxdepth", "pick_mode")] + [
        ("hmc_eps", C.c_double), ("nuts_delta", C.c_double), ("gj_tab", _dp)]


class State(C.Structure):
    _fields_ = [("X", _dp), ("lnL", _dp), ("lp", _dp), ("temp_of", _ip), ("slot_of", _ip), ("Ut", _dp), ("S", _dp),
                ("DE", _dp), ("AM", _dp), ("nacc", _up), ("jstat", _up), ("gj", _dp), ("AMflag", _up)]


class Replay(C.Structure):
    _fields_ = [("kinds", C.POINTER(C.c_uint8)), ("vals", _dp), ("bounds", C.POINTER(C.c_int64)),
                ("n", C.c_int64), ("pos", C.c_int64), ("err", C.c_int64)]


def build(force=False):
    src = os.path.join(_HERE, "ptmcmc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_log.restype = L.orc_exp.restype = L.orc_cos2pi.restype = C.c_double
        L.orc_log.argtypes = L.orc_exp.argtypes = L.orc_cos2pi.argtypes = [C.c_double]
        L.orc_log_n.restype = None
        L.orc_log_n.argtypes = [C.c_int64, _dp, _dp]
        L.orc_normal.restype = L.orc_normal_sin.restype = L.orc_sin2pi.restype = C.c_double
        L.orc_normal.argtypes = L.orc_normal_sin.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_unit_log.restype = C.c_double
        L.orc_unit_log.argtypes = [C.c_uint64]
        L.orc_unit_sincos64.restype = L.orc_unit_sincos32.restype = L.orc_unit_normals.restype = None
        L.orc_unit_sincos64.argtypes = [C.c_uint64, _dp, _dp]
        L.orc_unit_sincos32.argtypes = [C.c_uint32, _dp, _dp]
        L.orc_unit_normals.argtypes = [C.c_uint64, C.c_uint64, _dp, _dp]
        L.orc_unit_normal32.restype = C.c_double
        L.orc_unit_normal32.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_sin2pi.argtypes = [C.c_double]
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64]
        L.orc_index.restype = C.c_uint64
        L.orc_index.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_mh_steps.argtypes = [C.POINTER(Cfg), C.POINTER(State), C.c_int64, C.c_int, C.POINTER(Replay)]
        L.orc_swap_sweep.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int64, C.c_uint64, C.c_int, _ip, _up,
                                     C.POINTER(Replay)]
        L.orc_swap_oddeven.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int64, C.c_uint64, C.c_int, C.c_int, _ip, _up]
        L.orc_swap_oddeven.restype = None
        L.orc_swap_apply.argtypes = [C.POINTER(Cfg), C.POINTER(State), _ip, C.c_int64]
        L.orc_welford.argtypes = [C.c_int, C.c_int, C.c_int64, _dp, _dp, _dp, _dp]
        L.orc_welford2.argtypes = [C.c_int, C.c_int, C.c_int64, _dp, _dp, _dp, _dp, C.c_int]
        L.orc_pool_update.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_pool_update.restype = None
        L.orc_pool_update_rle.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _dp, _up, _dp, _dp, _dp]
        L.orc_pool_update_rle.restype = None
        L.orc_de_update.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.orc_de_update_pooled.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.orc_eval_state.argtypes = [C.POINTER(Cfg), C.POINTER(State)]
        L.orc_gradjump.argtypes = [C.POINTER(Cfg), C.c_int, _dp, C.c_int64, C.c_double, _dp, C.c_uint64,
                                   C.POINTER(Replay), _dp, _dp, C.POINTER(C.c_int64)]
        L.orc_logl_grad.restype = C.c_double
        L.orc_logl_grad.argtypes = [C.POINTER(Cfg), _dp, _dp]
        L.orc_logl.restype = C.c_double
        L.orc_logl.argtypes = [C.POINTER(Cfg), _dp]
        L.orc_eig_jacobi.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int]
        L.orc_eig_ql.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_div_by_count.argtypes = [C.c_long, _dp, _dp, _dp]
        assert L.orc_sizeof_cfg() == C.sizeof(Cfg)
        _lib = L
    return _lib


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


def dense_par(mu, P):
    """Parameter block of the dense likelihood: mu | Pt (Pt[j][i] = P[i][j], the gradient's table) | Tl (the half of the
    symmetric part of P the VALUE is summed over: Tl[k][i] = Ps[k][i] for k > i, Ps[i][i] / 2 for k == i, else 0)."""
    mu, P = np.asarray(mu, dtype=np.float64), np.asarray(P, dtype=np.float64)
    Pt = np.ascontiguousarray(P.T)
    Ps = (Pt + Pt.T) * 0.5
    Tl = np.tril(Ps, -1) + np.diag(np.diag(Ps) * 0.5)
    return np.concatenate([mu, Pt.ravel(), Tl.ravel()])


def interval_par(a, b, d):
    """Parameter block of the interval family (LOGL_INTERVAL): a | w = b - a | log w."""
    a = np.broadcast_to(np.asarray(a, dtype=np.float64), (d,))
    b = np.broadcast_to(np.asarray(b, dtype=np.float64), (d,))
    w = b - a
    return np.ascontiguousarray(np.concatenate([a, w, np.log(w)]))


def lanes_for(ndim, grad=False):
    """Lanes that share one chain in the HIP kernels (fixes the summation order); grad: with NUTS / HMC in the cycle."""
    if grad:
        return 4 if ndim <= 32 else (16 if ndim <= 112 else 64)
    return 4 if ndim <= 104 else (16 if ndim <= 416 else 64)


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return [int(v) for v in out]


def temperature_ladder(nchain, ndim, Tmin=1, Tmax=None):
    """PTMCMCSampler.py:699-720 (host-side; same numpy expressions)."""
    if nchain > 1:
        if Tmax is None:
            tstep = 1 + np.sqrt(2 / ndim)
        else:
            tstep = np.exp(np.log(Tmax / Tmin) / (nchain - 1))
        ladder = np.zeros(nchain)
        for ii in range(nchain):
            ladder[ii] = Tmin * tstep**ii
        return ladder
    return np.array([1])


JACOBI_MAX_SWEEPS = 30


def eig_jacobi(cov, max_sweeps=JACOBI_MAX_SWEEPS):
    """(Ut, S, sweeps) of a symmetric positive semi-definite matrix by the engine's Jacobi definition: eigenvectors as
    ROWS of Ut, eigenvalues descending."""
    cov = np.ascontiguousarray(cov, dtype=np.float64)
    d = len(cov)
    Ut, S = np.zeros((d, d)), np.zeros(d)
    n = lib().orc_eig_jacobi(d, _p(cov), _p(Ut), _p(S), max_sweeps)
    return Ut, S, n


def eig_ql(cov):
    """(Ut, S, iterations) by the engine's tridiagonal QL definition (orc_eig_ql): eigenvectors as ROWS of Ut, |eigenvalues| descending."""
    cov = np.ascontiguousarray(cov, dtype=np.float64)
    d = len(cov)
    Ut, S = np.zeros((d, d)), np.zeros(d)
    n = lib().orc_eig_ql(d, _p(cov), _p(Ut), _p(S))
    assert n >= 0, "QL did not converge"
    return Ut, S, n


def div_by_count(a, n):
    """a / n the way welford_rows_kernel forms it (reciprocal + two remainder corrections): must equal a / n exactly."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    n = np.ascontiguousarray(n, dtype=np.float64)
    out = np.empty_like(a)
    lib().orc_div_by_count(len(a), _p(a), _p(n), _p(out))
    return out


def welford(AM, mu, M2, it, fused=False):
    """In-place PTMCMCSampler.py:769-794; returns cov.  ``fused``: one fma per element (pooled mode)."""
    mem, d = AM.shape
    cov = np.empty((d, d))
    AMc = np.ascontiguousarray(AM)
    lib().orc_welford2(d, mem, it, _p(AMc), _p(mu), _p(M2), _p(cov), int(fused))
    return cov


def pool_slab(nwalkers, ndim):
    """Walkers per slab of the pooled statistics (the engine's rule, ptmi_abi.hip pool_slab): up to 512 slabs for the
    one-macro-tile shapes (ndim <= 111), up to 32 beyond (a slab's partial matrix is ndim x (ndim + 1) doubles)."""
    target = 512 if ndim + 1 <= 112 else 32
    return max(1, -(-nwalkers // target))


def pool_update(AM, mu, M2, it, slab=None):
    """Pooled-covariance epoch (orc_pool_update): AM [W][mem][d]; mu [d], M2 [d][d] updated in place; returns cov."""
    W, mem, d = AM.shape
    cov = np.empty((d, d))
    AMc = np.ascontiguousarray(AM)
    lib().orc_pool_update(d, W, mem, it, pool_slab(W, d) if slab is None else slab, _p(AMc), _p(mu), _p(M2), _p(cov))
    return cov


def pool_update_rle(AM, flag, mu, M2, it, slab=None):
    """Pooled-covariance epoch over run-length-compacted rows (orc_pool_update_rle; the engine's am_mode "rle"): flag [W][mem]
    uint64, bit 0 NEW, bit 1 KEY; every stored row once, weighted by its run length."""
    W, mem, d = AM.shape
    cov = np.empty((d, d))
    AMc = np.ascontiguousarray(AM)
    fl = np.ascontiguousarray(flag, dtype=np.uint64)
    assert fl.shape == (W, mem) and (fl[:, 0] & 3).all(), "ring row 0 of every walker is a stored row"
    lib().orc_pool_update_rle(d, W, mem, it, pool_slab(W, d) if slab is None else slab, _p(AMc), _p(fl, _up), _p(mu), _p(M2), _p(cov))
    return cov


def de_update(DE, AM):
    AMc = np.ascontiguousarray(AM)
    lib().orc_de_update(DE.shape[1], DE.shape[0], AMc.shape[0], _p(DE), _p(AMc))


def swap_sweep(ladder, lnL_pos, it=0, seed=0, walker0=0, uniforms=None):
    """Returns (map[W][n], acc[W][n]) for PTMCMCSampler.py:672-681."""
    lnL_pos = np.ascontiguousarray(np.atleast_2d(lnL_pos), dtype=np.float64)
    W, n = lnL_pos.shape
    ladder = np.ascontiguousarray(ladder, dtype=np.float64)
    m = np.zeros((W, n), dtype=np.int32)
    acc = np.zeros((W, n), dtype=np.uint64)
    rp = None
    if uniforms is not None:
        rp = make_replay(np.ones(len(uniforms), np.uint8), np.asarray(uniforms, float), np.zeros(len(uniforms), np.int64))
    err = lib().orc_swap_sweep(W, n, _p(ladder), _p(lnL_pos), it, seed, walker0, _p(m, _ip), _p(acc, _up),
                               C.byref(rp[0]) if rp else None)
    assert err == 0, "replay error %d" % err
    return m, acc


def swap_parity(it, tskip):
    """Odd/even mode: swap epoch e = it / tskip tries the pairs (k, k+1) with k = e (mod 2)."""
    return int((it // tskip if tskip > 0 else it) & 1)


def swap_oddeven(ladder, lnL_pos, parity, it=0, seed=0, walker0=0):
    """Odd/even counterpart of swap_sweep (engine mode, not in the reference)."""
    lnL_pos = np.ascontiguousarray(np.atleast_2d(lnL_pos), dtype=np.float64)
    W, n = lnL_pos.shape
    ladder = np.ascontiguousarray(ladder, dtype=np.float64)
    m = np.zeros((W, n), dtype=np.int32)
    acc = np.zeros((W, n), dtype=np.uint64)
    lib().orc_swap_oddeven(W, n, _p(ladder), _p(lnL_pos), it, seed, walker0, parity, _p(m, _ip), _p(acc, _up))
    return m, acc


def gj_tables(cov):
    """Whitening tables of the gradient jumps from L = cholesky(cov) (nutsjump.py:53-54), in the layout the kernels
    and the oracle read: [backward L, forward L^-1, gradient L^T], each used as out[i] = sum_k T[k][i] v[k]."""
    import scipy.linalg as sl
    cov = np.asarray(cov, dtype=np.float64)
    L = sl.cholesky(cov, lower=True)
    Li = sl.solve_triangular(L, np.eye(len(cov)), trans=0, lower=True)
    return np.ascontiguousarray(np.stack([L, Li, L.T]))


def gj_state():
    st = np.zeros(GJ_NSTATE)
    st[GJ_EPSBAR] = 1.0
    return st


def gradjump(kind, x, it, beta, state, cov, logl=("iso",), logp=("flat",), nburn=100, delta=0.6, hmc=(0.1, 2, 300),
             maxdepth=12, lanes=None, seed=0, sid=0, replay=None):
    """One NUTS / HMC jump call on the point x through the oracle (state: gj_state(), updated in place).
    replay = (kinds, vals, bounds) of the reference's global np.random draws."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    d = len(x)
    par_l, par_p = np.zeros(1), np.zeros(1)
    if logl[0] == "dense":
        par_l = dense_par(logl[1], logl[2])
    if logl[0] == "interval":
        par_l = interval_par(logl[1], logl[2], d)
    if logp[0] == "box":
        par_p = np.concatenate([np.asarray(logp[1], float), np.asarray(logp[2], float)])
    tab = gj_tables(cov)
    cfg = Cfg(ndim=d, ntemps=1, nwalkers=1, lanes=lanes_for(d) if lanes is None else lanes, logl_kind=LOGL[logl[0]],
              logp_kind=LOGP[logp[0]], seed=seed, logl_par=_p(par_l), logp_par=_p(par_p), gj_nburn=nburn,
              hmc_eps=hmc[0], hmc_min=hmc[1], hmc_max=hmc[2], nuts_maxdepth=maxdepth, nuts_delta=delta, gj_tab=_p(tab))
    q, qxy, nleap = np.zeros(d), np.zeros(1), C.c_int64(0)
    rp = keep = None
    if replay is not None:
        rp, keep = make_replay(*replay)
    err = lib().orc_gradjump(C.byref(cfg), J[kind], _p(x), it, beta, _p(state), sid, C.byref(rp) if rp is not None else None,
                             _p(q), _p(qxy), C.byref(nleap))
    assert err == 0, "replay error %d" % err
    if rp is not None:
        assert rp.pos == rp.n, "unused draws: %d of %d" % (rp.pos, rp.n)
    return q, float(qxy[0]), int(nleap.value)


def make_replay(kinds, vals, bounds):
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    bounds = np.ascontiguousarray(bounds, dtype=np.int64)
    r = Replay(_p(kinds, C.POINTER(C.c_uint8)), _p(vals), _p(bounds, C.POINTER(C.c_int64)), len(kinds), 0, 0)
    return r, (kinds, vals, bounds)


class OracleEngine(object):
    """All temperatures x walkers of the reference's sampler loop, on the CPU.

    Walker = an independent replica of the whole reference run (own ladder copy,
    own RNG streams; own cov/U/S/DE history when ``cov_mode='per_walker'``).
    """

    def __init__(self, ndim, ntemps, nwalkers, cov0, ladder=None, logl=("iso",), logp=("flat",),
                 weights=(20, 20, 20), cov_update=1000, burn=10000, tskip=100, seed=0,
                 cov_mode="per_walker", hot_chain=False, lanes=None, Tmin=1, Tmax=None,
                 ntemps_global=None, temp0=0, walker0=0, groups=None, swap_mode="sweep",
                 grad_weights=(0, 0), hmc=(0.1, 2, 300), nuts_delta=0.6, nuts_maxdepth=24, pick_mode="chain",
                 eig_mode="lapack", am_mode="auto", eig_lag=0):
        assert eig_lag >= 0
        # the engine's eig_lag: the factorization of a covariance epoch takes effect eig_lag segments late (pooled covariance, one
        # parameter group; whichever eigensolver: the host's LAPACK, or the restated device ones)
        self.eig_lag = int(eig_lag) if (groups is None and (cov_mode == "pooled" or eig_mode == "ql")) else 0
        self._eig_pending, self._eig_wait = False, 0
        assert swap_mode in ("sweep", "oddeven") and pick_mode in ("chain", "walker") and eig_mode in ("lapack", "jacobi", "ql")
        assert am_mode in ("auto", "rows", "rle")
        # the engine's am_mode: "rle" (pooled covariance on the block that holds rank 0) weights the pooled statistics by run lengths
        self.am_rle = am_mode != "rows" and cov_mode == "pooled" and temp0 == 0
        assert self.am_rle or am_mode != "rle"
        self.eig_mode = eig_mode
        self.pick_mode = pick_mode
        self.swap_mode = swap_mode
        self.d, self.nt, self.W = ndim, ntemps, nwalkers
        self.ntg = ntemps if ntemps_global is None else ntemps_global
        self.temp0, self.walker0 = temp0, walker0
        self.ladder = np.asarray(temperature_ladder(self.ntg, ndim, Tmin, Tmax) if ladder is None else ladder,
                                 dtype=np.float64)
        self.temps_mh = self.ladder[temp0:temp0 + ntemps].copy()
        if hot_chain and temp0 + ntemps == self.ntg:
            self.temps_mh[-1] = 1e80                       # PTMCMCSampler.py:281-282
        self.beta = 1 / self.temps_mh
        self.cov_update, self.burn, self.tskip, self.seed = cov_update, burn, tskip, seed
        self.per_walker = cov_mode == "per_walker"
        self.Wc = nwalkers if self.per_walker else 1
        self.lanes = lanes_for(ndim, grad=sum(grad_weights) > 0 or logl[0] == "interval") if lanes is None else lanes
        d, nt, W = ndim, ntemps, nwalkers
        self.X = np.zeros((W, nt, d))
        self.lnL = np.zeros((W, nt))
        self.lp = np.zeros((W, nt))
        self.temp_of = np.tile(np.arange(nt, dtype=np.int32), (W, 1))
        self.slot_of = self.temp_of.copy()
        self.groups = [np.arange(d)] if groups is None else [np.asarray(g, dtype=np.int64) for g in groups]
        self.ngr = len(self.groups)
        self.gsize = np.asarray([len(g) for g in self.groups], dtype=np.int32)
        self.gmask = np.zeros((self.ngr, d))
        for gi, g in enumerate(self.groups):
            self.gmask[gi, g] = 1.0
        self.cov = np.tile(np.asarray(cov0, dtype=np.float64), (self.Wc, 1, 1))
        self.Ut = np.zeros((self.Wc, self.ngr, d, d))
        self.S = np.zeros((self.Wc, self.ngr, d))
        for w in range(self.Wc):
            self._svd(w)
        self._initial_done = True                          # the initial factorization is the host's in every mode (PT:139-145)
        # adaptation state: per walker, or ONE pooled (mu, M2) in pooled mode
        self.mu = np.zeros((self.Wc, d))
        self.M2 = np.zeros((self.Wc, d, d))
        self.DE = np.zeros((self.Wc, burn, d))
        self.AM = np.zeros((W, cov_update, d))
        self.AMflag = np.full((W, cov_update), 2, dtype=np.uint64)      # every row starts as a KEY row
        self.nacc = np.zeros((W, nt), dtype=np.uint64)
        self.jstat = np.zeros((W, nt, J_NTYPES, 2), dtype=np.uint64)
        # gradient jumps (PTMCMCSampler.py:225-258): whitening from the INITIAL covariance, never adapted (nutsjump.py:45)
        self.grad_weights = tuple(int(v) for v in grad_weights)
        self.gj_tab = gj_tables(cov0) if sum(self.grad_weights) else np.zeros(1)
        self.gj = np.zeros((W, nt, GJ_NSTATE))
        self.gj[..., GJ_EPSBAR] = 1.0
        self.nswap = np.zeros((W, self.ntg), dtype=np.uint64)
        self.swap_proposed = 0
        self._par_l = np.zeros(1)
        self._par_p = np.zeros(1)
        if logl[0] == "dense":
            self._par_l = dense_par(logl[1], logl[2])
        if logl[0] == "interval":
            self._par_l = interval_par(logl[1], logl[2], d)
        if logp[0] == "box":
            self._par_p = np.concatenate([np.asarray(logp[1], float), np.asarray(logp[2], float)])
        self.cfg = Cfg(ndim=d, ntemps=nt, nwalkers=W, lanes=self.lanes, logl_kind=LOGL[logl[0]],
                       logp_kind=LOGP[logp[0]], w_scam=weights[0], w_am=weights[1], w_de=weights[2], de_on=0,
                       de_size=burn, cov_update=cov_update, tskip=tskip, cov_per_walker=int(self.per_walker),
                       ntemps_global=self.ntg, temp0=temp0, walker0=walker0, ngroups=self.ngr, seed=seed,
                       logl_par=_p(self._par_l), logp_par=_p(self._par_p), temps_mh=_p(self.temps_mh),
                       beta=_p(self.beta), gsize=_p(self.gsize, _ip), gmask=_p(self.gmask),
                       w_nuts=self.grad_weights[0], w_hmc=self.grad_weights[1], gj_nburn=burn, hmc_eps=hmc[0],
                       hmc_min=hmc[1], hmc_max=hmc[2], nuts_maxdepth=nuts_maxdepth, nuts_delta=nuts_delta,
                       gj_tab=_p(self.gj_tab), pick_mode={"chain": 0, "walker": 1}[pick_mode])
        self.iter = 0

    # -- helpers
    def _state(self):
        return State(_p(self.X), _p(self.lnL), _p(self.lp), _p(self.temp_of, _ip), _p(self.slot_of, _ip), _p(self.Ut),
                     _p(self.S), _p(self.DE), _p(self.AM) if self.temp0 == 0 else None, _p(self.nacc, _up),
                     _p(self.jstat, _up), _p(self.gj), _p(self.AMflag, _up) if (self.temp0 == 0 and self.am_rle) else None)

    def _svd(self, w):
        # LAPACK results depend on the BLAS thread count in the last bits, so the checker uses the product's rule:
        # per-walker mode = the reference's np.linalg.svd on one thread (PTMCMCSampler.py:145, 803); pooled mode (not a
        # replica of a reference run) = np.linalg.eigh, eigenvalues by decreasing size and in absolute value, on one
        # thread up to 256 parameters and on 8 beyond
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:
            import contextlib
            threadpool_limits = lambda limits: contextlib.nullcontext()   # noqa: E731
        if self.eig_mode in ("jacobi", "ql") and getattr(self, "_initial_done", False):
            if self.ngr > 1 or len(self.groups[0]) != self.d or not np.array_equal(self.groups[0], np.arange(self.d)):
                # parameter groups with the device QL solver (ptmi_eig_ql): one factorization per group's block of the covariance
                # (PTMCMCSampler.py:797-803), the block taken in ASCENDING parameter order, its vectors embedded in the full space
                assert self.eig_mode == "ql"
                for gi, g in enumerate(self.groups):
                    gs = np.sort(g)
                    Ut, S, _ = eig_ql(np.ascontiguousarray(self.cov[w][np.ix_(gs, gs)]))
                    self.Ut[w, gi] = 0.0
                    self.S[w, gi] = 0.0
                    self.Ut[w, gi][np.ix_(np.arange(len(gs)), gs)] = Ut
                    self.S[w, gi, :len(gs)] = S
                return
            Ut, S, _ = eig_jacobi(self.cov[w]) if self.eig_mode == "jacobi" else eig_ql(self.cov[w])   # the engine's device eigensolvers (covariance epochs only)
            self.Ut[w, 0], self.S[w, 0] = Ut, S
            return
        for gi, g in enumerate(self.groups):               # per group, PTMCMCSampler.py:139-145, 797-803
            c = self.cov[w][np.ix_(g, g)]
            if self.per_walker:
                with threadpool_limits(limits=1):
                    U, S, _ = np.linalg.svd(c)
            else:
                with threadpool_limits(limits=1 if len(c) <= 256 else 8):
                    ev, V = np.linalg.eigh(c)
                U, S = np.ascontiguousarray(V[:, ::-1]), np.abs(ev[::-1])
            self.set_eig(U, S, w, gi)

    def set_eig(self, U, S, w=0, gi=0):
        """Embed a group's eigenvectors (columns of U) in the full space, one per row."""
        g = self.groups[gi]
        self.Ut[w, gi] = 0.0
        self.S[w, gi] = 0.0
        self.Ut[w, gi][np.ix_(np.arange(len(g)), g)] = np.asarray(U).T
        self.S[w, gi, :len(g)] = S

    def init_state(self, p0):
        p0 = np.asarray(p0, dtype=np.float64)
        self.X[...] = p0 if p0.ndim == 3 else np.broadcast_to(p0, self.X.shape)
        lib().orc_eval_state(C.byref(self.cfg), C.byref(self._state()))
        if self.temp0 == 0:
            self.AM[:, 0, :] = self.X[np.arange(self.W), self.slot_of[:, 0]]   # updateChains(p0, ..., 0), :491
            self.AMflag[:, 0] = 2

    def by_temp(self, a):
        """Reorder a per-slot array [W][nt][...] into temperature order."""
        return np.take_along_axis(a, self.slot_of.reshape(self.slot_of.shape + (1,) * (a.ndim - 2)).astype(np.int64), 1)

    def lnprob(self):
        return self.beta[self.temp_of] * self.lnL + self.lp

    # -- epochs (PTMCMCSampler.py:545-585)
    def _epochs(self, it):
        cu, burn = self.cov_update, self.burn
        if self.temp0 == 0:
            if (it - 1) % cu == 0 and it - 1 != 0:
                self._eig_finish()                             # a factorization still pending from the epoch before
                if self.per_walker:
                    for w in range(self.W):
                        self.cov[w] = welford(self.AM[w], self.mu[w], self.M2[w], it - 1)
                elif self.am_rle:
                    self.cov[0] = pool_update_rle(self.AM, self.AMflag, self.mu[0], self.M2[0], it - 1)
                else:
                    self.cov[0] = pool_update(self.AM, self.mu[0], self.M2[0], it - 1)
                if self.eig_lag:
                    self._eig_pending, self._eig_wait = True, self.eig_lag
                else:
                    for w in range(self.Wc):
                        self._svd(w)
            if (it - 1) % burn == 0 and it - 1 != 0:
                if self.per_walker:
                    for w in range(self.W):
                        de_update(self.DE[w], self.AM[w])
                else:
                    lib().orc_de_update_pooled(self.d, burn, cu, self.W, _p(self.DE), _p(self.AM))
        if it - 1 == burn:
            self.cfg.de_on = 1

    def _eig_finish(self):
        if self._eig_pending:
            for w in range(self.Wc):
                self._svd(w)
        self._eig_pending, self._eig_wait = False, 0

    def _next_event(self, it, niter):
        """Last iteration of the segment that starts at ``it`` (no epoch inside, swap only at its end)."""
        end = niter
        for per in (self.cov_update, self.burn) + ((self.tskip,) if self.tskip > 0 and self.ntg > 1 else ()):
            end = min(end, ((it - 1) // per + 1) * per)
        return end

    def swap(self, it, replay0=None):
        """Single-process PT swap over the local ranks (requires the whole ladder local)."""
        assert self.nt == self.ntg
        lnL_pos = np.ascontiguousarray(self.by_temp(self.lnL))
        m = np.zeros((self.W, self.nt), dtype=np.int32)
        if self.swap_mode == "oddeven":
            assert replay0 is None
            lib().orc_swap_oddeven(self.W, self.nt, _p(self.ladder), _p(lnL_pos), it, self.seed, self.walker0,
                                   swap_parity(it, self.tskip), _p(m, _ip), _p(self.nswap, _up))
        else:
            err = lib().orc_swap_sweep(self.W, self.nt, _p(self.ladder), _p(lnL_pos), it, self.seed, self.walker0,
                                       _p(m, _ip), _p(self.nswap, _up), replay0)
            assert err == 0, "replay error %d" % err
        lib().orc_swap_apply(C.byref(self.cfg), C.byref(self._state()), _p(m, _ip), it)
        self.swap_proposed += 1
        return m

    def run(self, niter, replay=None, record=False):
        """Advance ``niter`` iterations. ``replay``: list (one per rank) of (kinds, vals, bounds)."""
        rp_arr, keep = None, []
        if replay is not None:
            assert self.W == 1
            rp_arr = (Replay * self.nt)()
            for t, (k, v, b) in enumerate(replay):
                r, ka = make_replay(k, v, b)
                keep.append(ka)
                rp_arr[t] = r
        rec = None
        if record:
            rec = dict(X=[self.by_temp(self.X).copy()], lnL=[self.by_temp(self.lnL).copy()],
                       lnprob=[self.by_temp(self.lnprob()).copy()])
        last = self.iter + niter
        it = self.iter + 1
        while it <= last:
            self._epochs(it)
            end = it if record else self._next_event(it, last)
            err = lib().orc_mh_steps(C.byref(self.cfg), C.byref(self._state()), it, end - it + 1, rp_arr)
            assert err == 0, "replay error %d at iter %d" % (err, it)
            if self.tskip > 0 and self.ntg > 1 and end % self.tskip == 0:
                self.swap(end, C.byref(rp_arr[0]) if rp_arr is not None else None)
            if self._eig_pending:                              # eig_lag segments after the epoch its table takes effect
                self._eig_wait -= 1
                if self._eig_wait <= 0:
                    self._eig_finish()
            if record:
                rec["X"].append(self.by_temp(self.X).copy())
                rec["lnL"].append(self.by_temp(self.lnL).copy())
                rec["lnprob"].append(self.by_temp(self.lnprob()).copy())
            it = end + 1
        self.iter = last
        if rp_arr is not None:
            self.replay_left = [int(rp_arr[t].n - rp_arr[t].pos) for t in range(self.nt)]
        if record:
            return {k: np.asarray(v) for k, v in rec.items()}
        return None
