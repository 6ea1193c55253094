"""Host-side gradient jumps against the reference's own outputs (nutsjump.py), same global numpy seed."""
import contextlib
import io

import numpy as np
import pytest


@pytest.mark.parametrize("tag", ["nuts", "nuts_forced", "hmc"])
def test_gradient_jumps_reproduce_the_reference(golden, tag):
    from ptmcmcsampler_amd.gradjump import HMCJump, NUTSJump
    g = golden("gradjump")
    P, cov = g["P"], g["cov"]

    def ll_grad(x):
        return -0.5 * np.dot(x, np.dot(P, x)), -np.dot(P, x)

    def lp_grad(x):
        return 0.0, np.zeros_like(x)

    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        j = {"nuts": lambda: NUTSJump(ll_grad, lp_grad, cov, nburn=25, delta=0.6),
             "nuts_forced": lambda: NUTSJump(ll_grad, lp_grad, cov, nburn=10, force_trajlen=5, force_epsilon=0.3),
             "hmc": lambda: HMCJump(ll_grad, lp_grad, cov, nburn=25, stepsize=0.15, nminsteps=2, nmaxsteps=20)}[tag]()
    assert "WARNING: GradientJumps not yet adaptive" in out.getvalue()          # the reference prints this too
    assert j.__name__ == str(g[tag + "_name"])
    np.random.seed(2024)
    xs = g[tag + "_x"]
    for it in range(1, len(g[tag + "_q"]) + 1):
        beta = 1.0 if it % 7 else 0.4
        q, qxy = j(xs[it - 1], it, beta)
        assert np.array_equal(np.asarray(q), g[tag + "_q"][it - 1]), (tag, it)
        assert qxy == g[tag + "_qxy"][it - 1], (tag, it)
        assert (j.epsilon if j.epsilon is not None else -1.0) == g[tag + "_eps"][it - 1], (tag, it)


def test_nuts_targets_the_distribution():
    """Statistical sanity of the NUTS jump inside a plain Metropolis loop (always accepted by construction)."""
    from ptmcmcsampler_amd.gradjump import NUTSJump
    rs = np.random.RandomState(3)
    d = 3
    A = rs.randn(d, d)
    C = A @ A.T / d + 0.5 * np.eye(d)
    P = np.linalg.inv(C)
    with contextlib.redirect_stdout(io.StringIO()):
        j = NUTSJump(lambda x: (-0.5 * x @ P @ x, -P @ x), lambda x: (0.0, np.zeros_like(x)), np.eye(d), nburn=200)
    np.random.seed(5)
    x = np.zeros(d)
    xs = []
    for it in range(1, 1501):
        x, qxy = j(x, it, 1.0)
        xs.append(x)
    xs = np.asarray(xs[300:])
    assert np.max(np.abs(np.cov(xs.T) - C)) / np.max(C) < 0.25


class _RecordGlobalDraws(object):
    """Records the global np.random draws of one jump call as an oracle replay (kinds, values, bounds)."""

    def __init__(self, orc):
        self.orc, self.k, self.v, self.b = orc, [], [], []

    def _put(self, kind, vals, bound=0):
        for x in np.atleast_1d(vals):
            self.k.append(kind)
            self.v.append(float(x))
            self.b.append(bound)

    def __enter__(self):
        o = self.o = (np.random.randn, np.random.exponential, np.random.uniform, np.random.randint)
        orc = self.orc

        def randn(*a):
            r = o[0](*a)
            self._put(orc.K_NRM, r)
            return r

        def exponential(*a, **kw):
            r = o[1](*a, **kw)
            self._put(orc.K_EXP, r)
            return r

        def uniform(*a, **kw):
            r = o[2](*a, **kw)
            self._put(orc.K_UNI, r)
            return r

        def randint(lo, hi=None, *a, **kw):
            r = o[3](lo, hi, *a, **kw)
            self._put(orc.K_INT, r, hi if hi is not None else lo)
            return r

        np.random.randn, np.random.exponential, np.random.uniform, np.random.randint = randn, exponential, uniform, randint
        return self

    def __exit__(self, *exc):
        np.random.randn, np.random.exponential, np.random.uniform, np.random.randint = self.o


@pytest.mark.parametrize("tag", ["nuts", "hmc"])
def test_oracle_gradient_jumps_replay_the_reference(golden, tag):
    """The C oracle's NUTS / HMC (what the device kernels are checked against) fed the reference's own draws: the
    draws are recorded from the host restatement, which the test above pins to the reference bit for bit."""
    from oracle import oracle as orc
    from ptmcmcsampler_amd.gradjump import HMCJump, NUTSJump
    g = golden("gradjump")
    P, cov = g["P"], g["cov"]
    d = len(P)

    def ll_grad(x):
        return -0.5 * np.dot(x, np.dot(P, x)), -np.dot(P, x)

    def lp_grad(x):
        return 0.0, np.zeros_like(x)

    with contextlib.redirect_stdout(io.StringIO()):
        j = NUTSJump(ll_grad, lp_grad, cov, nburn=25, delta=0.6) if tag == "nuts" else \
            HMCJump(ll_grad, lp_grad, cov, nburn=25, stepsize=0.15, nminsteps=2, nmaxsteps=20)
    kw = dict(nburn=25) if tag == "nuts" else dict(nburn=25, hmc=(0.15, 2, 20))
    np.random.seed(2024)
    xs, st, leaps = g[tag + "_x"], orc.gj_state(), 0
    for it in range(1, len(g[tag + "_q"]) + 1):
        beta = 1.0 if it % 7 else 0.4
        with _RecordGlobalDraws(orc) as rec:
            q, _ = j(xs[it - 1], it, beta)
        assert np.array_equal(q, g[tag + "_q"][it - 1])
        qo, qxyo, nl = orc.gradjump(tag, xs[it - 1], it, beta, st, cov, logl=("dense", np.zeros(d), P),
                                    replay=(rec.k, rec.v, rec.b), lanes=4, **kw)      # asserts every draw is consumed, in kind
        leaps += nl
        np.testing.assert_allclose(qo, g[tag + "_q"][it - 1], rtol=0, atol=1e-10)
        assert abs(qxyo - g[tag + "_qxy"][it - 1]) < 1e-10
        if tag == "nuts":
            assert abs(st[orc.GJ_EPS] - g["nuts_eps"][it - 1]) < 1e-10 * g["nuts_eps"][it - 1]
    assert leaps > (100 if tag == "nuts" else 20)


def _interval_callbacks(a, b):
    """The reference's own NUTS workload (tests/test_nuts.py: GaussianLikelihood :13-47 inside intervalTransform :50-140), written out
    with its operations in its order: what a user hands to the sampler as logl_grad / logp_grad."""
    def ll_grad(p):
        x = (b - a) * np.exp(p) / (1 + np.exp(p)) + a
        ll = -0.5 * np.sum(x**2) - len(x) * 0.5 * np.log(2 * np.pi)
        lj = np.sum(np.log(b - a) + p - 2 * np.log(1.0 + np.exp(p)))
        dxdp = (b - a) * np.exp(p) / (1 + np.exp(p)) ** 2
        return ll + lj, -x * dxdp + (1 - np.exp(p)) / (1 + np.exp(p))

    def lp_grad(p):
        x = (b - a) * np.exp(p) / (1 + np.exp(p)) + a
        return (0.0 if np.all(a <= x) and np.all(b >= x) else -np.inf), np.zeros_like(p)

    return ll_grad, lp_grad


@pytest.mark.parametrize("tag", ["nuts", "hmc"])
def test_interval_family_replays_the_reference_nuts_workload(golden, tag):
    """``("interval", a, b)`` (include/ptmi.h PTMI_LOGL_INTERVAL) against the reference's jump objects run on the reference's own test
    likelihood (tests/golden/make_golden.py gen_interval): the host restatement reproduces the reference's proposals bit for bit, and
    the C oracle -- the definition the device kernels are compared with -- fed the same draws lands on them to 1e-9 (its exp / log
    are its own, its sums run in the kernels' lane order)."""
    from oracle import oracle as orc
    from ptmcmcsampler_amd.gradjump import HMCJump, NUTSJump
    g = golden("interval")
    a, b, cov = g["j_a"], g["j_b"], g["j_cov"]
    ll_grad, lp_grad = _interval_callbacks(a, b)
    with contextlib.redirect_stdout(io.StringIO()):
        j = NUTSJump(ll_grad, lp_grad, cov, nburn=25, delta=0.6) if tag == "nuts" else \
            HMCJump(ll_grad, lp_grad, cov, nburn=25, stepsize=0.2, nminsteps=2, nmaxsteps=12)
    kw = dict(nburn=25) if tag == "nuts" else dict(nburn=25, hmc=(0.2, 2, 12))
    np.random.seed(515)
    xs, st, leaps = g[tag + "_x"], orc.gj_state(), 0
    for it in range(1, len(g[tag + "_q"]) + 1):
        beta = 1.0 if it % 5 else 0.5
        with _RecordGlobalDraws(orc) as rec:
            q, qxy = j(xs[it - 1], it, beta)
        assert np.array_equal(q, g[tag + "_q"][it - 1]), (tag, it)
        assert qxy == g[tag + "_qxy"][it - 1]
        qo, qxyo, nl = orc.gradjump(tag, xs[it - 1], it, beta, st, cov, logl=("interval", a, b),
                                    replay=(rec.k, rec.v, rec.b), lanes=4, **kw)      # asserts every draw is consumed, in kind
        leaps += nl
        np.testing.assert_allclose(qo, g[tag + "_q"][it - 1], rtol=0, atol=1e-9)
        assert abs(qxyo - g[tag + "_qxy"][it - 1]) < 1e-9
        if tag == "nuts":
            assert abs(st[orc.GJ_EPS] - g["nuts_eps"][it - 1]) < 1e-9 * g["nuts_eps"][it - 1]
    assert leaps > (100 if tag == "nuts" else 20)
