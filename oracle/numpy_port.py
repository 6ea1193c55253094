"""Reference-equivalent CPU baseline -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

A one-chain-at-a-time NumPy port with the same per-iteration structure and the same
interpreter-level costs as the reference's ``PTMCMCOneStep``
(PTMCMCSampler/PTMCMCSampler.py:530-629): Python-callback likelihood, NumPy SCAM / AM /
DE proposals (:820-985), Python-loop Welford update every ``covUpdate`` (:769-803, the
reference's dominant cost) and ``np.linalg.svd``.  ``bench.py`` times it on the GPU
box's host cores as ``cpu_baseline`` (kind "port"): the reference itself cannot travel.
``oracle/baseline_calibration.json`` records how its per-core rate compares with the
real reference measured in the build container (must agree within +-20 %).
"""
import time

import numpy as np


class _Comm(object):
    """The reference pays for two communicator calls per iteration even on one rank (:501, :523)."""

    def barrier(self):
        pass

    def bcast(self, obj, root=0):
        return obj


class _Bound(object):
    """logl / logp reach the sampler through a wrapper object that splices args and kwargs in (:1072-1086)."""

    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, args, kwargs

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


class ChainPort(object):
    """One chain with the reference's call graph: run -> one_step -> _jump -> proposal(x, iter, beta) -> wrapped
    callbacks -> update_chains, every piece paying the interpreter costs its counterpart pays."""

    def __init__(self, ndim, logl, logp, cov, temp=1.0, covUpdate=1000, burn=10000, weights=(20, 20, 20), seed=0):
        self.ndim, self.logl, self.logp = ndim, _Bound(logl, [], {}), _Bound(logp, [], {})
        self.temp = np.float64(temp)                 # a ladder entry, i.e. a NumPy scalar (:278)
        self.cov = np.array(cov, dtype=float)
        self.groups = [np.arange(0, ndim)]           # parameters move through an index array (:129-131)
        self.U, self.S = [[]], [[]]
        self.U[0], self.S[0], _ = np.linalg.svd(self.cov)
        self.M2, self.mu = np.zeros((ndim, ndim)), np.zeros(ndim)
        self.covUpdate, self.burn, self.rank, self.nchain = covUpdate, burn, 0, 1
        self.AM = np.zeros((covUpdate, ndim))
        self.DE = np.zeros((burn, ndim))
        self.stream = np.random.default_rng(seed)
        self.comm = _Comm()
        self.cycle, self.aux, self.stats = [], [], {}
        self.wde = weights[2]
        self.naccepted = 0
        for f, w in ((self.scam, weights[0]), (self.am, weights[1])):
            self.add(f, w)

    def add(self, f, weight):
        self.cycle += [f] * weight
        if weight:
            self.stats.setdefault(f.__name__, [0, 0])

    def _size(self):
        """(group index, size, jump scale): the common head of SCAM and AM (:838-862, :896-920)."""
        gi = self.stream.integers(0, len(self.groups))
        nd = len(self.groups[gi])
        prob = self.stream.random()
        if prob > 0.97:
            scale = 10
        elif prob > 0.9:
            scale = 0.2
        else:
            scale = 1.0
        if self.temp <= 100:
            scale *= np.sqrt(self.temp)
        return gi, nd, scale

    def scam(self, x, iter, beta):
        q, qxy = x.copy(), 0
        gi, nd, scale = self._size()
        ind = np.unique(self.stream.integers(0, nd, 1))
        cd = 2.4 / np.sqrt(2 * len(ind)) * scale
        q[self.groups[gi]] += self.stream.standard_normal() * cd * np.sqrt(self.S[gi][ind]) * self.U[gi][:, ind].flatten()
        return q, qxy

    def am(self, x, iter, beta):
        q, qxy = x.copy(), 0
        gi, nd, scale = self._size()
        y = np.dot(self.U[gi].T, x[self.groups[gi]])
        ind = np.arange(len(self.groups[gi]))
        cd = 2.4 / np.sqrt(2 * len(ind)) * scale
        y[ind] = y[ind] + self.stream.standard_normal(len(ind)) * cd * np.sqrt(self.S[gi][ind])
        q[self.groups[gi]] = np.dot(self.U[gi], y)
        return q, qxy

    def de(self, x, iter, beta):
        q, qxy = x.copy(), 0
        gi = self.stream.integers(0, len(self.groups))
        nd = len(self.groups[gi])
        n = len(self.DE)
        mm, nn = self.stream.integers(0, n), self.stream.integers(0, n)
        while mm == nn:
            nn = self.stream.integers(0, n)
        if self.stream.random() > 0.5:
            scale = 1.0
        else:
            scale = self.stream.random() * 2.4 / np.sqrt(2 * nd) * np.sqrt(1 / beta)
        for ii in range(nd):
            q[self.groups[gi][ii]] += scale * (self.DE[mm, self.groups[gi][ii]] - self.DE[nn, self.groups[gi][ii]])
        return q, qxy

    def _jump(self, x, iter):
        k = self.stream.integers(0, len(self.cycle))
        q, qxy = self.cycle[k](x, iter, 1 / self.temp)
        if len(self.aux) > 0:
            for aux in self.aux:
                q, extra = aux(x, q, iter, 1 / self.temp)
                qxy += extra
        return q, qxy, self.cycle[k].__name__

    def update_recursive(self, it_done, mem):
        it = it_done - mem
        if it == 0:
            self.M2 = np.zeros((self.ndim, self.ndim))
            self.mu = np.zeros(self.ndim)
        for ii in range(mem):
            diff = np.zeros(self.ndim)
            it += 1
            for jj in range(self.ndim):
                diff[jj] = self.AM[ii, jj] - self.mu[jj]
                self.mu[jj] += diff[jj] / it
            self.M2 += np.outer(diff, (self.AM[ii, :] - self.mu))
        self.cov[:, :] = self.M2 / (it - 1)
        for ct, group in enumerate(self.groups):
            covgroup = np.zeros((len(group), len(group)))
            for ii in range(len(group)):
                for jj in range(len(group)):
                    covgroup[ii, jj] = self.cov[group[ii], group[jj]]
            self.U[ct], self.S[ct], _ = np.linalg.svd(covgroup)

    def update_chains(self, x, lnl, lnp, it):
        if self.rank == 0:
            self.AM[it % self.covUpdate, :] = x
        if it % self.thin == 0:
            ind = int(it / self.thin)
            self._chain[ind, :] = x
            self._lnlike[ind] = lnl
            self._lnprob[ind] = lnp
        if it % self.isave == 0 and it > 0:
            for ind in range((it - self.isave) // self.thin + 1, it // self.thin + 1):
                self.sink.write("\t".join(["%22.22f" % (self._chain[ind, kk]) for kk in range(self.ndim)]))
                self.sink.write("\t%f\t%f\t%f\t%f\n" % (self._lnprob[ind], self._lnlike[ind], self.naccepted / it, 1))

    def one_step(self, x, lnl, lnp, it):
        """PTMCMCOneStep (:530-629) for a lone rank."""
        if self.rank == 0 and (it - 1) % self.covUpdate == 0 and (it - 1) != 0:
            self.update_recursive(it - 1, self.covUpdate)
        if self.rank == 0 and (it - 1) % self.burn == 0 and (it - 1) != 0:
            self.DE = np.concatenate([self.DE[self.covUpdate:], self.AM])
        if self.rank == 0 and (it - 1) == self.burn and self.wde:
            self.add(self.de, self.wde)
        y, qxy, name = self._jump(x, it)
        self.stats[name][0] += 1
        lp = self.logp(y)
        if lp == float(-np.inf):
            newp = -np.inf
        else:
            newl = self.logl(y)
            newp = 1 / self.temp * newl + lp
        diff = newp - lnp + qxy
        if diff > np.log(self.stream.random()):
            x, lnl, lnp = y, newl, newp
            self.naccepted += 1
            self.stats[name][1] += 1
        if self.nchain > 1 and it % 100 == 0:
            pass
        self.update_chains(x, lnl, lnp, it)
        return x, lnl, lnp

    def run(self, p0, niter, thin=10, isave=1000):
        import io
        self.thin, self.isave, self.sink = thin, isave, io.StringIO()
        x = np.array(p0, dtype=float)
        lp = self.logp(x)
        lnl = self.logl(x)
        lnp = 1 / self.temp * lnl + lp
        N = int(niter / thin) + 1
        self._chain, self._lnlike, self._lnprob = np.zeros((N, self.ndim)), np.zeros(N), np.zeros(N)
        it, done = 0, False
        while not done:
            it += 1
            self.comm.barrier()                            # :501
            x, lnl, lnp = self.one_step(x, lnl, lnp, it)
            if self.rank == 0 and it >= niter:
                done = True
            done = self.comm.bcast(done, root=0)           # :523
        return x


def iso_logl(x):
    return -0.5 * np.sum(x ** 2)


def flat_logp(x):
    return 0.0


def _worker(args):
    ndim, temp, niter, covUpdate, burn, weights, seed = args
    c = ChainPort(ndim, iso_logl, flat_logp, np.eye(ndim) * 0.01, temp, covUpdate, burn, weights, seed)
    t0 = time.perf_counter()
    c.run(np.zeros(ndim), niter)
    return time.perf_counter() - t0


def usable_cores():
    """Cores this process may actually use: the smaller of the visible CPUs, the affinity mask and the cgroup CPU quota
    (a container can show 256 CPUs and be throttled to 16)."""
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
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def time_baseline(ndim=100, niter=6000, covUpdate=1000, burn=10000, weights=(20, 0, 0), cores=None, ladder=None):
    """One chain per process on ``cores`` host cores (the reference's one-chain-per-rank model; default: every core
    this process may use); returns (updates per second over all cores, cores, description)."""
    import multiprocessing as mp
    cores = cores or usable_cores()
    if ladder is None:
        ladder = (1 + np.sqrt(2 / ndim)) ** np.arange(cores)
    jobs = [(ndim, float(ladder[r % len(ladder)]), niter, covUpdate, burn, weights, 100 + r) for r in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        times = pool.map(_worker, jobs, chunksize=1)
    # the chains run side by side; the slowest one's own loop time is the wall time of the sampling itself (process
    # start-up and imports, which a long reference run amortizes, are left out)
    wall = max(times)
    return cores * niter / wall, cores, "%d chains (one per core) x %d iterations, ndim=%d, covUpdate=%d; slowest chain %.1f s" % (
        cores, niter, ndim, covUpdate, wall)
