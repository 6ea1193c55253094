#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the real reference.

Runs ONLY in the build container (it imports /root/reference, which never
travels to the GPU box).  The committed ``*.npz`` files are data: inputs, the
reference's recorded RNG draws and the reference's outputs.  Nothing of the
reference's source is stored.

Recipe: stub ``PTMCMCSampler.version``, import the package
(falls back to its size-1 dummy comm), replace ``sampler.stream`` with a
recording proxy, and for multi-temperature runs give every rank (one thread
each) an in-process fake communicator implementing the eight duck-typed
methods the reference calls.

Usage:  python tests/golden/make_golden.py
"""
import copy
import os
import queue
import sys
import tempfile
import threading
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# draw kinds in the recorded stream
K_INT, K_UNI, K_NRM, K_SHUF = 0, 1, 2, 3


def import_reference():
    sys.path.insert(0, REF)
    ver = types.ModuleType("PTMCMCSampler.version")
    ver.version = "0+ref"
    sys.modules["PTMCMCSampler.version"] = ver
    from PTMCMCSampler import PTMCMCSampler as PT  # noqa

    return PT


class RecordingStream(object):
    """Proxy around a numpy Generator that logs every draw, in order."""

    def __init__(self, gen):
        self.gen = gen
        self.kinds, self.vals, self.bounds = [], [], []

    def _log(self, kind, vals, bound=0):
        for v in np.atleast_1d(vals):
            self.kinds.append(kind)
            self.vals.append(float(v))
            self.bounds.append(int(bound))

    def integers(self, low, high=None, size=None):
        r = self.gen.integers(low, high, size)
        self._log(K_INT, r, high if high is not None else low)
        return r

    def random(self, size=None):
        r = self.gen.random(size)
        self._log(K_UNI, r)
        return r

    def uniform(self, *a, **k):
        r = self.gen.uniform(*a, **k)
        self._log(K_UNI, r)
        return r

    def standard_normal(self, size=None):
        r = self.gen.standard_normal(size)
        self._log(K_NRM, r)
        return r

    def shuffle(self, arr):
        self.gen.shuffle(arr)
        self._log(K_SHUF, 0.0, len(arr))

    def arrays(self):
        return (
            np.asarray(self.kinds, dtype=np.uint8),
            np.asarray(self.vals, dtype=np.float64),
            np.asarray(self.bounds, dtype=np.int64),
        )


class ForcedUniform(object):
    """Generator stand-in whose random() returns a fixed value (to hit the rare
    scale branches of the proposals); everything else comes from the real one."""

    def __init__(self, gen, value):
        self.gen, self.value = gen, value

    def random(self, size=None):
        return self.value

    def __getattr__(self, name):
        return getattr(self.gen, name)


class World(object):
    def __init__(self, n):
        self.n = n
        self.bar = threading.Barrier(n)
        self.slots = [None] * n
        self.shared = None
        self.queues = {}
        self.lock = threading.Lock()

    def q(self, src, dst, tag):
        with self.lock:
            return self.queues.setdefault((src, dst, tag), queue.Queue())


class ThreadComm(object):
    """The duck-typed comm of nompi4py.py:6-33, for n in-process ranks."""

    def __init__(self, world, rank):
        self.w, self.r = world, rank

    def Get_rank(self):
        return self.r

    def Get_size(self):
        return self.w.n

    def barrier(self):
        self.w.bar.wait()

    def send(self, obj, dest=1, tag=55):
        self.w.q(self.r, dest, tag).put(copy.deepcopy(obj))

    def recv(self, source=1, tag=55):
        return self.w.q(source, self.r, tag).get()

    def gather(self, obj, root=0):
        self.w.slots[self.r] = copy.deepcopy(obj)
        self.w.bar.wait()
        out = list(self.w.slots) if self.r == root else None
        self.w.bar.wait()
        return out

    def scatter(self, lst, root=0):
        if self.r == root:
            self.w.shared = [copy.deepcopy(v) for v in lst]
        self.w.bar.wait()
        out = self.w.shared[self.r]
        self.w.bar.wait()
        return out

    def bcast(self, obj, root=0):
        if self.r == root:
            self.w.shared = obj
        self.w.bar.wait()
        out = self.w.shared
        self.w.bar.wait()
        return out


class RootOnlyComm(object):
    """Single rank-0 object that sees prepared gather() results; used to drive
    the reference's PTswap root code without threads."""

    def __init__(self, n, gathers):
        self.n, self.gathers, self.scattered = n, list(gathers), []

    def Get_rank(self):
        return 0

    def Get_size(self):
        return self.n

    def barrier(self):
        pass

    def gather(self, obj, root=0):
        return self.gathers.pop(0)

    def scatter(self, lst, root=0):
        if lst is None:
            return None
        self.scattered.append(copy.deepcopy(lst))
        return lst[0]

    def bcast(self, obj, root=0):
        return obj

    def send(self, *a, **k):
        pass

    def recv(self, *a, **k):
        pass


# ---------------------------------------------------------------- workloads
def iso_logl(x):
    return -0.5 * np.sum(x**2)


def flat_logp(x):
    return 0.0


class Box(object):
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def __call__(self, x):
        if np.all(self.lo <= x) and np.all(self.hi >= x):
            return 0.0
        return -np.inf


class Dense(object):
    """-(x-mu)^T P (x-mu) / 2, as tests/test_simple.py:14-41 builds it."""

    def __init__(self, mu, icov):
        self.mu, self.icov = mu, icov

    def __call__(self, x):
        diff = x - self.mu
        return -np.dot(diff, np.dot(self.icov, diff)) / 2.0


def make_dense(rs, ndim, pmin, pmax):
    mu = rs.uniform(pmin, pmax, ndim)
    cov = 0.5 - rs.rand(ndim**2).reshape((ndim, ndim))
    cov = np.triu(cov)
    cov += cov.T - np.diag(cov.diagonal())
    cov = np.dot(cov, cov)
    return mu, np.linalg.inv(cov)


# ---------------------------------------------------------------- fixtures
def gen_ladder(PT, out):
    cases, res = [], {}
    for i, (n, d, Tmin, Tmax) in enumerate(
        [(1, 5, 1, None), (2, 2, 1, None), (4, 6, 1, None), (64, 100, 1, None), (8, 20, 1.0, 50.0), (5, 3, 2.0, None)]
    ):
        s = PT.PTSampler.__new__(PT.PTSampler)
        s.nchain, s.ndim = n, d
        res["ladder_%d" % i] = np.asarray(s.temperatureLadder(Tmin, Tmax=Tmax), dtype=np.float64)
        cases.append([n, d, Tmin, -1.0 if Tmax is None else Tmax])
    res["cases"] = np.asarray(cases, dtype=np.float64)
    np.savez(os.path.join(out, "ladder.npz"), **res)


def bare_sampler(PT, ndim, cov, tmp, seed=11):
    s = PT.PTSampler(ndim, iso_logl, flat_logp, np.copy(cov), outDir=tmp, verbose=False, seed=seed)
    s.stream = RecordingStream(s.stream)
    return s


def gen_proposals(PT, out, tmp):
    rs = np.random.RandomState(5)
    res, meta = {}, []
    ci = 0
    for d in (5, 20, 100):
        A = rs.randn(d, d)
        cov = A @ A.T / d + 0.1 * np.eye(d)
        s = bare_sampler(PT, d, cov, tmp)
        s._DEbuffer = rs.randn(37, d)
        for temp in (1.0, 3.7, 100.0, 101.0, 1e80):
            s.temp = temp
            for kind, fn in ((0, s.covarianceJumpProposalSCAM), (1, s.covarianceJumpProposalAM), (2, s.DEJump)):
                real = s.stream.gen
                forced = [None] * (6 if d < 100 else 2) + [0.5, 0.93, 0.985, 0.2]
                for rep, fu in enumerate(forced):
                    x = rs.randn(d)
                    s.stream.kinds, s.stream.vals, s.stream.bounds = [], [], []
                    s.stream.gen = real if fu is None else ForcedUniform(real, fu)
                    q, qxy = fn(x, 1, 1.0 / temp)
                    s.stream.gen = real
                    k, v, b = s.stream.arrays()
                    res["x_%d" % ci], res["q_%d" % ci] = x, np.asarray(q)
                    res["dk_%d" % ci], res["dv_%d" % ci], res["db_%d" % ci] = k, v, b
                    meta.append([d, temp, kind, qxy])
                    ci += 1
        res["U_d%d" % d], res["S_d%d" % d], res["DE_d%d" % d] = s.U[0], s.S[0], s._DEbuffer
    res["meta"] = np.asarray(meta, dtype=np.float64)
    np.savez_compressed(os.path.join(out, "proposals.npz"), **res)


def gen_welford(PT, out, tmp):
    rs = np.random.RandomState(6)
    res = {}
    for d, mem in ((5, 50), (100, 40)):
        s = bare_sampler(PT, d, np.eye(d) * 0.01, tmp)
        s.covUpdate = mem
        L = rs.randn(d, d) * 0.3 + np.eye(d)
        for ep in range(3):
            s._AMbuffer = rs.randn(mem, d) @ L.T + 0.5
            s._updateRecursive((ep + 1) * mem, mem)
            tag = "d%d_e%d" % (d, ep)
            res["am_" + tag] = s._AMbuffer.copy()
            res["mu_" + tag], res["M2_" + tag], res["cov_" + tag] = s.mu.copy(), s.M2.copy(), s.cov.copy()
            res["U_" + tag], res["S_" + tag] = s.U[0].copy(), s.S[0].copy()
    np.savez_compressed(os.path.join(out, "welford.npz"), **res)


def gen_debuffer(PT, out, tmp):
    rs = np.random.RandomState(7)
    d, mem, burn = 4, 10, 35
    s = bare_sampler(PT, d, np.eye(d), tmp)
    s._DEbuffer = np.zeros((burn, d))
    res = {"shape": np.asarray([d, mem, burn])}
    for ep in range(5):
        s._AMbuffer = rs.randn(mem, d)
        s._updateDEbuffer((ep + 1) * burn, burn)
        res["am_%d" % ep], res["de_%d" % ep] = s._AMbuffer.copy(), s._DEbuffer.copy()
    np.savez_compressed(os.path.join(out, "debuffer.npz"), **res)


def gen_ptswap(PT, out, tmp):
    rs = np.random.RandomState(8)
    res, meta = {}, []
    ci = 0
    for n, d, spread in ((2, 3, 1.0), (4, 6, 3.0), (64, 100, 8.0), (64, 100, 60.0), (16, 5, 0.0), (8, 2, np.inf), (512, 4, 40.0)):
        for rep in range(3):
            if np.isinf(spread):  # -inf likelihoods (out-of-prior starts) among the ranks
                lnL = -np.abs(rs.randn(n)) * 5
                lnL[rs.rand(n) < 0.4] = -np.inf
            else:
                lnL = -0.5 * d - spread * np.abs(rs.randn(n)) * np.linspace(1, 3, n)
            p0s = [rs.randn(d) for _ in range(n)]
            comm = RootOnlyComm(n, [list(lnL), p0s])
            s = PT.PTSampler(d, iso_logl, flat_logp, np.eye(d), comm=comm, outDir=tmp, verbose=False, seed=100 + ci)
            s.stream = RecordingStream(s.stream)
            s.initialize(10)
            s.stream.kinds, s.stream.vals, s.stream.bounds = [], [], []
            s.PTswap(p0s[0], lnL[0], 0.0, 10)
            new_p0s, new_lnL, acc = comm.scattered[-3:]
            k, v, b = s.stream.arrays()
            res["lnL_%d" % ci], res["ladder_%d" % ci] = lnL, np.asarray(s.ladder, dtype=np.float64)
            res["u_%d" % ci] = v
            res["p0s_%d" % ci], res["newp0s_%d" % ci] = np.asarray(p0s), np.asarray(new_p0s)
            res["newlnL_%d" % ci], res["acc_%d" % ci] = np.asarray(new_lnL), np.asarray(acc)
            meta.append([n, d])
            ci += 1
    res["meta"] = np.asarray(meta)
    np.savez_compressed(os.path.join(out, "ptswap.npz"), **res)


def run_traj(PT, name, out, ndim, nranks, logl, logp, p0, cov0, sample_kw, seed, extra=None, hot=False, groups=None):
    """Full sample() run with per-rank recorded draws and per-epoch cov snapshots."""
    tmp = tempfile.mkdtemp()
    world = World(nranks)
    samplers = [None] * nranks
    epochs = []
    errs = []

    def rank_main(r):
        try:
            comm = ThreadComm(world, r) if nranks > 1 else None
            kw = dict(outDir=tmp, verbose=False, seed=seed)
            if groups is not None:
                kw["groups"] = [np.asarray(g) for g in groups]
            if comm is not None:
                kw["comm"] = comm
            s = PT.PTSampler(ndim, logl, logp, np.copy(cov0), **kw)
            s.stream = RecordingStream(s.stream)
            samplers[r] = s
            if r == 0:
                orig = s._updateRecursive

                def snap(it, mem):
                    orig(it, mem)
                    epochs.append((it, s.mu.copy(), s.M2.copy(), s.cov.copy(), s.U[0].copy(), s.S[0].copy()))

                s._updateRecursive = snap
            s.sample(np.copy(p0), hotChain=hot, **sample_kw)
        except BaseException as e:  # noqa
            errs.append(e)
            world.bar.abort()
            raise

    if nranks == 1:
        rank_main(0)
    else:
        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
        [t.start() for t in th]
        [t.join() for t in th]
    if errs:
        raise errs[0]

    res = dict(extra or {})
    if groups is not None:
        res["groups_flat"] = np.concatenate([np.asarray(g) for g in groups])
        res["groups_size"] = np.asarray([len(g) for g in groups])
    res["ndim"], res["nranks"], res["seed"] = ndim, nranks, seed
    res["p0"], res["cov0"] = p0, cov0
    for k, v in sample_kw.items():
        res["kw_" + k] = v
    res["hot"] = int(hot)
    res["ladder"] = np.asarray(samplers[0].ladder, dtype=np.float64)
    res["temps"] = np.asarray([s.temp for s in samplers], dtype=np.float64)
    for r, s in enumerate(samplers):
        k, v, b = s.stream.arrays()
        res["dk_%d" % r], res["dv_%d" % r], res["db_%d" % r] = k, v, b
        res["chain_%d" % r], res["lnlike_%d" % r], res["lnprob_%d" % r] = s._chain, s._lnlike, s._lnprob
        res["nacc_%d" % r] = s.naccepted
        res["nswap_%d" % r] = s.nswap_accepted
        res["swapprop_%d" % r] = s.swapProposed
        names = sorted(s.jumpDict)
        res["jnames_%d" % r] = np.asarray(names)
        res["jstats_%d" % r] = np.asarray([s.jumpDict[n] for n in names], dtype=np.int64)
    res["nepochs"] = len(epochs)
    for i, (it, mu, M2, cov, U, S) in enumerate(epochs):
        res["ep_it_%d" % i], res["ep_mu_%d" % i], res["ep_M2_%d" % i] = it, mu, M2
        res["ep_cov_%d" % i], res["ep_U_%d" % i], res["ep_S_%d" % i] = cov, U, S
    # the chain file the reference wrote for rank 0 (format pin for the writer row)
    f0 = samplers[0].fname
    with open(f0) as fh:
        lines = fh.read().splitlines()
    res["chainfile_name"] = os.path.basename(f0)
    res["chainfile_head"] = np.asarray(lines[:3])
    res["chainfile_nlines"] = len(lines)
    np.savez_compressed(os.path.join(out, name + ".npz"), **res)
    print(name, "ranks", nranks, "epochs", len(epochs), "acc", [s.naccepted for s in samplers])


def gen_trajectories(PT, out):
    rs = np.random.RandomState(9)
    # T1: single chain, adaptation on, DE enters after burn
    d = 5
    run_traj(PT, "traj_single_d5", out, d, 1, iso_logl, flat_logp, rs.randn(d) * 0.3, np.eye(d) * 0.01,
             dict(Niter=600, covUpdate=100, burn=200, thin=1, isave=100, Tskip=100,
                  SCAMweight=20, AMweight=20, DEweight=20), seed=4242)
    # T2: box prior with wide proposals -> the lp=-inf short-circuit fires often
    d = 4
    lo, hi = -np.ones(d), np.ones(d)
    run_traj(PT, "traj_single_box_d4", out, d, 1, iso_logl, Box(lo, hi), rs.uniform(-0.5, 0.5, d), np.eye(d) * 0.5,
             dict(Niter=400, covUpdate=100, burn=150, thin=1, isave=100, Tskip=100,
                  SCAMweight=30, AMweight=15, DEweight=10), seed=77, extra=dict(box_lo=lo, box_hi=hi))
    # T3: four temperatures, swaps every 10
    d = 6
    run_traj(PT, "traj_pt4_d6", out, d, 4, iso_logl, flat_logp, rs.randn(d) * 0.2, np.eye(d) * 0.01,
             dict(Niter=300, covUpdate=50, burn=100, thin=1, isave=50, Tskip=10,
                  SCAMweight=20, AMweight=20, DEweight=20), seed=1234)
    # T4: dense Gaussian + box prior, 3 temperatures, hot chain, thin>1
    d = 8
    mu, icov = make_dense(rs, d, 0.0, 10.0)
    lo, hi = np.zeros(d), 10.0 * np.ones(d)
    run_traj(PT, "traj_pt3_dense_d8", out, d, 3, Dense(mu, icov), Box(lo, hi), rs.uniform(0, 10, d), np.eye(d) * 0.01,
             dict(Niter=240, covUpdate=40, burn=80, thin=2, isave=40, Tskip=8,
                  SCAMweight=20, AMweight=20, DEweight=20), seed=31337, hot=True,
             extra=dict(box_lo=lo, box_hi=hi, dense_mu=mu, dense_icov=icov))
    # T6/T7: parameter groups (per-group SVD and group-restricted jumps, PTMCMCSampler.py:129-145, 839, 897, 955)
    d = 6
    run_traj(PT, "traj_groups_d6", out, d, 1, iso_logl, flat_logp, rs.randn(d) * 0.3, np.eye(d) * 0.02,
             dict(Niter=500, covUpdate=100, burn=200, thin=1, isave=100, Tskip=100,
                  SCAMweight=20, AMweight=20, DEweight=20), seed=606, groups=[[0, 1, 2], [3, 4, 5]])
    d = 5
    run_traj(PT, "traj_groups_pt2_d5", out, d, 2, iso_logl, flat_logp, rs.randn(d) * 0.3, np.eye(d) * 0.02,
             dict(Niter=300, covUpdate=50, burn=100, thin=1, isave=50, Tskip=10,
                  SCAMweight=20, AMweight=20, DEweight=20), seed=707, groups=[[0, 1, 2, 3, 4], [3, 1], [2]])
    # T5: the SCAM-only slice at the bench dimension (d=100), 2 temperatures, adaptation off
    d = 100
    run_traj(PT, "traj_pt2_scam_d100", out, d, 2, iso_logl, flat_logp, np.zeros(d), np.eye(d) * 0.01,
             dict(Niter=200, covUpdate=1000, burn=10000, thin=1, isave=100, Tskip=20,
                  SCAMweight=20, AMweight=0, DEweight=0), seed=1234)


def gen_gradjump(out):
    """HMC and NUTS jumps of the reference (nutsjump.py) on a dense Gaussian, global np.random seeded."""
    import contextlib
    import io
    from PTMCMCSampler.nutsjump import HMCJump, MALAJump, NUTSJump
    rs = np.random.RandomState(12)
    d = 5
    A = rs.randn(d, d)
    P = np.linalg.inv(A @ A.T / d + 0.3 * np.eye(d))
    cov = np.linalg.inv(P) * 0.8

    def ll_grad(x):
        return -0.5 * np.dot(x, np.dot(P, x)), -np.dot(P, x)

    def lp_grad(x):
        return 0.0, np.zeros_like(x)

    res = dict(P=P, cov=cov)
    for tag, make, ncall in (("nuts", lambda: NUTSJump(ll_grad, lp_grad, cov, nburn=25, delta=0.6), 45),
                             ("nuts_forced", lambda: NUTSJump(ll_grad, lp_grad, cov, nburn=10, force_trajlen=5, force_epsilon=0.3), 12),
                             ("mala", lambda: MALAJump(ll_grad, lp_grad, cov, nburn=25), 20),
                             ("hmc", lambda: HMCJump(ll_grad, lp_grad, cov, nburn=25, stepsize=0.15, nminsteps=2, nmaxsteps=20), 30)):
        with contextlib.redirect_stdout(io.StringIO()):
            j = make()
        np.random.seed(2024)
        x = rs.randn(d)
        xs, qs, qxys, eps = [x.copy()], [], [], []
        for it in range(1, ncall + 1):
            beta = 1.0 if it % 7 else 0.4
            q, qxy = j(x, it, beta)
            qs.append(np.array(q))
            qxys.append(float(qxy))
            eps.append(float(j.epsilon) if j.epsilon is not None else -1.0)
            if it % 3:                      # the sampler would accept or not; feed some proposals back
                x = np.array(q)
            xs.append(x.copy())
        res[tag + "_x"], res[tag + "_q"] = np.asarray(xs), np.asarray(qs)
        res[tag + "_qxy"], res[tag + "_eps"] = np.asarray(qxys), np.asarray(eps)
        res[tag + "_name"] = j.__name__
    np.savez_compressed(os.path.join(out, "gradjump.npz"), **res)



def reference_test_classes(path, names):
    """Classes of one of the reference's TEST modules, taken out of it by name and executed here (the module itself imports mpi4py,
    which this container does not have).  Only their outputs are stored."""
    import ast
    tree = ast.parse(open(path).read())
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def gen_interval(out):
    """The reference's own NUTS workload (tests/test_nuts.py): GaussianLikelihood inside intervalTransform.  Values and gradients of
    its lnlikefn_grad / lnpriorfn at sample points, and NUTS / HMC calls of the reference's jump objects on it (global np.random
    seeded, as gen_gradjump) -- what the ("interval", a, b) device family (include/ptmi.h PTMI_LOGL_INTERVAL) is pinned to."""
    import contextlib
    import io
    from PTMCMCSampler.nutsjump import HMCJump, NUTSJump
    Gauss, Interval = reference_test_classes(os.path.join(REF, "tests", "test_nuts.py"), ["GaussianLikelihood", "intervalTransform"])
    rs = np.random.RandomState(77)
    res = {}
    # (1) values: the test's own box (0, 10) at 40-d, and an uneven box at 7-d
    for tag, d, a, b in (("t40", 40, np.zeros(40), np.full(40, 10.0)), ("u7", 7, rs.uniform(-3, 0, 7), rs.uniform(0.5, 6, 7))):
        glo = Gauss(ndim=d, pmin=0.0, pmax=1.0)
        glo.a, glo.b = a.copy(), b.copy()
        glt = Interval(glo)
        P = np.concatenate([rs.randn(12, d) * 2.0, rs.randn(4, d) * 8.0])
        vals = [glt.lnlikefn_grad(p) for p in P]
        res[tag + "_a"], res[tag + "_b"], res[tag + "_p"] = a, b, P
        res[tag + "_ll"] = np.array([v[0] for v in vals])
        res[tag + "_grad"] = np.array([v[1] for v in vals])
        res[tag + "_lp"] = np.array([glt.lnpriorfn(p) for p in P])
        res[tag + "_x"] = np.array([glt.backward(p) for p in P])
    # (2) the jumps, 8-d on (0, 10), whitened by a diagonal covariance as the pair layout of the kernels wants it
    d = 8
    glo = Gauss(ndim=d, pmin=0.0, pmax=10.0)
    glt = Interval(glo)
    cov = np.diag(rs.uniform(0.5, 2.0, d))
    res["j_cov"], res["j_a"], res["j_b"] = cov, glo.a, glo.b
    for tag, make, ncall in (("nuts", lambda: NUTSJump(glt.lnlikefn_grad, glt.lnpriorfn_grad, cov, nburn=25, delta=0.6), 40),
                             ("hmc", lambda: HMCJump(glt.lnlikefn_grad, glt.lnpriorfn_grad, cov, nburn=25, stepsize=0.2, nminsteps=2, nmaxsteps=12), 25)):
        with contextlib.redirect_stdout(io.StringIO()):
            j = make()
        np.random.seed(515)
        x = rs.randn(d) * 0.5 - 1.0
        xs, qs, qxys, eps = [x.copy()], [], [], []
        for it in range(1, ncall + 1):
            beta = 1.0 if it % 5 else 0.5
            q, qxy = j(x, it, beta)
            qs.append(np.array(q))
            qxys.append(float(qxy))
            eps.append(float(j.epsilon) if j.epsilon is not None else -1.0)
            if it % 3:
                x = np.array(q)
            xs.append(x.copy())
        res[tag + "_x"], res[tag + "_q"] = np.asarray(xs), np.asarray(qs)
        res[tag + "_qxy"], res[tag + "_eps"] = np.asarray(qxys), np.asarray(eps)
    np.savez_compressed(os.path.join(out, "interval.npz"), **res)


def gen_config1(PT, out):
    """BASELINE configs[0]: the workload of the reference's examples/simple.py:52-122 -- 20-d dense Gaussian
    built from the global NumPy state, box prior [0, 10], cov0 = 0.01 I, the UniformJump custom proposal with weight 5,
    sample(p0, 10000, burn=500, thin=1, covUpdate=500, SCAM = AM = DE = 20) -- run by the reference with fixed seeds.
    Stored: the target, the start, and the statistics of the reference's chain."""
    tmp = tempfile.mkdtemp()
    np.random.seed(20240501)
    ndim, pmin, pmax = 20, 0.0, 10.0
    mu, icov = make_dense(np.random, ndim, pmin, pmax)
    logl, logp = Dense(mu, icov), Box(np.ones(ndim) * pmin, np.ones(ndim) * pmax)
    p0 = np.random.uniform(pmin, pmax, ndim)
    cov0 = np.eye(ndim) * 0.1**2
    def jump(x, it, beta):
        return np.random.uniform(pmin, pmax, len(x)), 0

    # 10000 iterations do not bring this 20-d truncated Gaussian to equilibrium (the mean log-likelihood is still rising):
    # runs of the reference itself differ by several posterior standard deviations.  Four of them give the yardstick.
    means, covs, accs, lls = [], [], [], []
    for seed in (42, 43, 44, 45):
        np.random.seed(seed)
        s = PT.PTSampler(ndim, logl, logp, np.copy(cov0), outDir=tmp, verbose=False, seed=seed)
        s.addProposalToCycle(jump, 5)
        s.sample(np.copy(p0), 10000, burn=500, thin=1, covUpdate=500, SCAMweight=20, AMweight=20, DEweight=20)
        chain = s._chain[2500:10001]
        means.append(chain.mean(0))
        covs.append(np.cov(chain, rowvar=False))
        accs.append(s.naccepted / 10000.0)
        lls.append(s._lnlike[2500:10001].mean())
    res = dict(mu=mu, icov=icov, p0=p0, cov0=cov0, ndim=ndim, pmin=pmin, pmax=pmax,
               ref_means=np.asarray(means), ref_covs=np.asarray(covs), ref_accs=np.asarray(accs), ref_lnlike_means=np.asarray(lls),
               jnames=np.asarray(sorted(s.jumpDict)), jstats=np.asarray([s.jumpDict[n] for n in sorted(s.jumpDict)], dtype=np.int64),
               jumps_txt=np.asarray(open(os.path.join(tmp, "jumps.txt")).read().splitlines()),
               nrows=len(open(s.fname).read().splitlines()), chainfile_name=os.path.basename(s.fname))
    np.savez_compressed(os.path.join(out, "config1.npz"), **res)


def gen_resume(PT, out):
    """A chain file the reference wrote (600 iterations, thin 2, isave 100), and what the reference does when it resumes
    from it (PTMCMCSampler.py:290-319, 591-599): the file rows are replayed as the chain's states, so the adaptive
    covariance and the DE history are rebuilt from them.  Stored: the file, the adaptation snapshots of the replay, the
    state at the iteration where sampling resumes."""
    tmp = tempfile.mkdtemp()
    d = 4
    rs = np.random.RandomState(9)
    mu, icov = make_dense(rs, d, 0.0, 10.0)
    logl, logp = Dense(mu, icov), Box(np.zeros(d), 10 * np.ones(d))
    p0, cov0 = rs.uniform(2, 8, d), np.eye(d) * 0.25
    kw = dict(thin=2, isave=100, covUpdate=100, burn=200, SCAMweight=20, AMweight=20, DEweight=20)
    s = PT.PTSampler(d, logl, logp, np.copy(cov0), outDir=tmp, verbose=False, seed=3)
    s.sample(np.copy(p0), 600, **kw)
    text = open(s.fname).read()
    s2 = PT.PTSampler(d, logl, logp, np.copy(cov0), outDir=tmp, verbose=False, seed=4, resume=True)
    epochs, orig = [], s2._updateRecursive

    def snap(it, mem):
        orig(it, mem)
        epochs.append((it, s2.mu.copy(), s2.M2.copy(), s2.cov.copy()))

    s2._updateRecursive = snap
    state = {}
    step = s2.PTMCMCOneStep

    def one(p, lnl, lnp, it):
        r = step(p, lnl, lnp, it)
        if it == s2.resumeLength * s2.thin - 1:            # the last replayed iteration
            state.update(x=np.array(r[0]), lnl=float(r[1]), lnp=float(r[2]), nacc=float(s2.naccepted), de=s2._DEbuffer.copy(),
                         am=s2._AMbuffer.copy())
        return r

    s2.PTMCMCOneStep = one
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        s2.sample(np.copy(p0), 1000, **kw)
    final = open(s2.fname).read().splitlines()
    res = dict(mu=mu, icov=icov, p0=p0, cov0=cov0, ndim=d, chainfile=np.asarray(text.splitlines()), chainfile_name=os.path.basename(s.fname),
               resume_length=s2.resumeLength, final_rows=len(final), replay_x=state["x"], replay_lnl=state["lnl"], replay_lnp=state["lnp"],
               replay_nacc=state["nacc"], replay_de=state["de"], replay_am=state["am"], nepochs=len(epochs))
    for k, v in kw.items():
        res["kw_" + k] = v
    for i, (it, m, M2, cov) in enumerate(epochs):
        res["ep_it_%d" % i], res["ep_mu_%d" % i], res["ep_M2_%d" % i], res["ep_cov_%d" % i] = it, m, M2, cov
    np.savez_compressed(os.path.join(out, "resume.npz"), **res)


def main():
    PT = import_reference()
    tmp = tempfile.mkdtemp()
    gen_ladder(PT, HERE)
    gen_proposals(PT, HERE, tmp)
    gen_welford(PT, HERE, tmp)
    gen_debuffer(PT, HERE, tmp)
    gen_ptswap(PT, HERE, tmp)
    gen_trajectories(PT, HERE)
    gen_gradjump(HERE)
    gen_interval(HERE)
    gen_config1(PT, HERE)
    gen_resume(PT, HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %8d B" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
