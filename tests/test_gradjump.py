"""Host-side gradient jumps against the reference's own outputs (nutsjump.py), same global numpy seed."""
import contextlib
import io

import numpy as np
import pytest


@pytest.mark.parametrize("tag", ["nuts", "nuts_forced", "hmc", "mala"])
def test_gradient_jumps_reproduce_the_reference(golden, tag):
    from ptmcmcsampler_amd.gradjump import HMCJump, MALAJump, NUTSJump
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
             "mala": lambda: MALAJump(ll_grad, lp_grad, cov, nburn=25),
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
