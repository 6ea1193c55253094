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


class ChainPort(object):
    def __init__(self, ndim, logl, logp, cov, temp=1.0, covUpdate=1000, burn=10000, weights=(20, 20, 20), seed=0):
        self.ndim, self.logl, self.logp, self.temp = ndim, logl, logp, temp
        self.cov = np.array(cov, dtype=float)
        self.U, self.S, _ = np.linalg.svd(self.cov)
        self.M2, self.mu = np.zeros((ndim, ndim)), np.zeros(ndim)
        self.covUpdate, self.burn = covUpdate, burn
        self.AM = np.zeros((covUpdate, ndim))
        self.DE = np.zeros((burn, ndim))
        self.rng = np.random.default_rng(seed)
        self.cycle = ["scam"] * weights[0] + ["am"] * weights[1]
        self.wde = weights[2]
        self.naccepted = 0
        self.jumps = dict(scam=[0, 0], am=[0, 0], de=[0, 0])

    # the reference also pays for these every iteration / every isave (:321-339, :501, :523, :741-745, :1085)
    def _call(self, f, x, *args, **kwargs):
        return f(x, *args, **kwargs)

    def _noop(self, *a, **k):
        return a[0] if a else None

    def _store(self, x, lnl, lnp, it, thin, isave, sink):
        if it % thin == 0:
            ind = int(it / thin)
            self._chain[ind, :] = x
            self._lnlike[ind] = lnl
            self._lnprob[ind] = lnp
        if it % isave == 0 and it > 0:
            for ind in range((it - isave) // thin + 1, it // thin + 1):
                sink.write("\t".join(["%22.22f" % (self._chain[ind, kk]) for kk in range(self.ndim)]))
                sink.write("\t%f\t%f\t%f\t%f\n" % (self._lnprob[ind], self._lnlike[ind], self.naccepted / it, 1))

    def _scale(self):
        prob = self.rng.random()
        scale = 10 if prob > 0.97 else (0.2 if prob > 0.9 else 1.0)
        if self.temp <= 100:
            scale *= np.sqrt(self.temp)
        return scale

    def scam(self, x):
        q = x.copy()
        self.rng.integers(0, 1)
        scale = self._scale()
        ind = np.unique(self.rng.integers(0, self.ndim, 1))
        cd = 2.4 / np.sqrt(2 * len(ind)) * scale
        q += self.rng.standard_normal() * cd * np.sqrt(self.S[ind]) * self.U[:, ind].flatten()
        return q

    def am(self, x):
        self.rng.integers(0, 1)
        scale = self._scale()
        y = np.dot(self.U.T, x)
        cd = 2.4 / np.sqrt(2 * self.ndim) * scale
        y = y + self.rng.standard_normal(self.ndim) * cd * np.sqrt(self.S)
        return np.dot(self.U, y)

    def de(self, x):
        q = x.copy()
        self.rng.integers(0, 1)
        n = len(self.DE)
        mm, nn = self.rng.integers(0, n), self.rng.integers(0, n)
        while mm == nn:
            nn = self.rng.integers(0, n)
        if self.rng.random() > 0.5:
            scale = 1.0
        else:
            scale = self.rng.random() * 2.4 / np.sqrt(2 * self.ndim) * np.sqrt(self.temp)
        for ii in range(self.ndim):
            q[ii] += scale * (self.DE[mm, ii] - self.DE[nn, ii])
        return q

    def update_recursive(self, it_done):
        it = it_done - self.covUpdate
        if it == 0:
            self.M2[:] = 0
            self.mu[:] = 0
        for ii in range(self.covUpdate):
            diff = np.zeros(self.ndim)
            it += 1
            for jj in range(self.ndim):
                diff[jj] = self.AM[ii, jj] - self.mu[jj]
                self.mu[jj] += diff[jj] / it
            self.M2 += np.outer(diff, (self.AM[ii, :] - self.mu))
        self.cov[:, :] = self.M2 / (it - 1)
        self.U, self.S, _ = np.linalg.svd(self.cov)

    def run(self, p0, niter, thin=10, isave=1000):
        import io
        x = np.array(p0, dtype=float)
        lnl = self.logl(x)
        lnp = lnl / self.temp + self.logp(x)
        fn = dict(scam=self.scam, am=self.am, de=self.de)
        N = int(niter / thin) + 1
        self._chain, self._lnlike, self._lnprob = np.zeros((N, self.ndim)), np.zeros(N), np.zeros(N)
        sink = io.StringIO()
        for it in range(1, niter + 1):
            self._noop()                                   # comm.barrier(), :501
            if (it - 1) % self.covUpdate == 0 and it - 1 != 0:
                self.update_recursive(it - 1)
            if (it - 1) % self.burn == 0 and it - 1 != 0:
                self.DE = np.concatenate([self.DE[self.covUpdate:], self.AM])
            if it - 1 == self.burn and self.wde:
                self.cycle = self.cycle + ["de"] * self.wde
            name = self.cycle[self.rng.integers(0, len(self.cycle))]
            y = fn[name](x)
            self.jumps[name][0] += 1
            lp = self._call(self.logp, y)
            if lp == -np.inf:
                newp = -np.inf
            else:
                newl = self._call(self.logl, y)
                newp = 1 / self.temp * newl + lp
            if newp - lnp + 0 > np.log(self.rng.random()):
                x, lnl, lnp = y, newl, newp
                self.naccepted += 1
                self.jumps[name][1] += 1
            self.AM[it % self.covUpdate, :] = x
            self._store(x, lnl, lnp, it, thin, isave, sink)
            self._noop(it >= niter)                        # comm.bcast(runComplete), :523
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
