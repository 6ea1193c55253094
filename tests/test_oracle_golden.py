"""The CPU oracle against the reference's own outputs (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference with recorded RNG draws)."""
import math

import numpy as np
import pytest

from oracle import oracle as orc

NAMES = ["covarianceJumpProposalSCAM", "covarianceJumpProposalAM", "DEJump"]


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert orc.philox([f, f, f, f], [f, f]) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def _ulps(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / math.ulp(b) if b != 0 else abs(a) / 5e-324


def test_math_close_to_libm():
    L = orc.lib()
    rs = np.random.RandomState(0)
    xs = np.concatenate([rs.rand(4000), 2.0 ** rs.uniform(-60, 0, 2000), [1.0, 2.0 ** -53, 0.5, 0.75, 1e-300, 3e-310]])
    assert max(_ulps(L.orc_log(float(x)), math.log(x)) for x in xs) <= 1.0
    assert L.orc_log(0.0) == -math.inf and math.isnan(L.orc_log(-1.0))
    es = np.concatenate([rs.uniform(-745, 709, 4000), rs.uniform(-2, 2, 2000), [0.0, -0.0, 709.78, -745.1, -800.0, 710.0]])
    for x in es:
        try:
            want = math.exp(x)
        except OverflowError:
            want = math.inf
        got = L.orc_exp(float(x))
        assert _ulps(got, want) <= (1.0 if want > 2.3e-308 else 2.0), x
    assert math.isnan(L.orc_exp(math.nan)) and L.orc_exp(-math.inf) == 0.0 and L.orc_exp(math.inf) == math.inf
    us = np.concatenate([rs.rand(6000), [0.0, 0.25, 0.5, 0.75, 0.125, 1 - 2.0 ** -53]])
    import mpmath
    mpmath.mp.dps = 40
    err = max(abs(L.orc_cos2pi(float(u)) - float(mpmath.cos(2 * mpmath.pi * mpmath.mpf(float(u))))) for u in us)
    assert err < 2.3e-16                                  # about one ulp at |cos| ~ 1
    assert L.orc_cos2pi(0.25) == 0.0 and L.orc_cos2pi(0.5) == -1.0 and L.orc_cos2pi(0.0) == 1.0
    err = max(abs(L.orc_sin2pi(float(u)) - float(mpmath.sin(2 * mpmath.pi * mpmath.mpf(float(u))))) for u in us)
    assert err < 2.3e-16
    assert L.orc_sin2pi(0.25) == 1.0 and L.orc_sin2pi(0.5) == 0.0 and L.orc_sin2pi(0.75) == -1.0


def test_normal_and_uniform_moments():
    L = orc.lib()
    rs = np.random.RandomState(1)
    w = rs.randint(0, 2 ** 63, size=(200000, 2)).astype(np.uint64) * np.uint64(2) + rs.randint(0, 2, (200000, 2)).astype(np.uint64)
    z = np.array([L.orc_normal(int(a), int(b)) for a, b in w[:50000]])
    zs = np.array([L.orc_normal_sin(int(a), int(b)) for a, b in w[:50000]])
    assert abs(zs.mean()) < 0.02 and abs(zs.std() - 1) < 0.02 and abs(np.corrcoef(z, zs)[0, 1]) < 0.02
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and abs((z ** 4).mean() - 3) < 0.15
    assert L.orc_uniform(2 ** 64 - 1) < 1.0 and L.orc_uniform(0) == 0.0
    assert L.orc_index(2 ** 64 - 1, 10) == 9 and L.orc_index(0, 10) == 0


def test_ladder(golden):
    g = golden("ladder")
    for i, (n, d, Tmin, Tmax) in enumerate(g["cases"]):
        got = orc.temperature_ladder(int(n), int(d), Tmin, None if Tmax < 0 else Tmax)
        assert np.array_equal(np.asarray(got, dtype=np.float64), g["ladder_%d" % i])


def test_proposals_match_reference(golden):
    """SCAM / AM / DE proposals (PTMCMCSampler.py:820-985) from the reference's recorded draws."""
    g = golden("proposals")
    meta = g["meta"]
    worst = {0: 0.0, 1: 0.0, 2: 0.0}
    for ci, (d, temp, kind, qxy) in enumerate(meta):
        d, kind = int(d), int(kind)
        x, q = g["x_%d" % ci], g["q_%d" % ci]
        e = orc.OracleEngine(d, 1, 1, np.eye(d), ladder=[temp], weights=(1, 1, 1), cov_update=4, burn=37, tskip=0)
        e.cfg.de_on = 1
        e.set_eig(g["U_d%d" % d], g["S_d%d" % d])
        e.DE[0] = g["DE_d%d" % d]
        e.init_state(x)
        # _jump's pick selects the proposal; a zero accept-uniform makes the step accept q
        k = np.concatenate([[0], g["dk_%d" % ci], [1]]).astype(np.uint8)
        v = np.concatenate([[float(kind)], g["dv_%d" % ci], [0.0]])
        b = np.concatenate([[3], g["db_%d" % ci], [0]]).astype(np.int64)
        e.run(1, replay=[(k, v, b)])
        assert e.replay_left == [0]
        assert qxy == 0
        err = np.max(np.abs(e.X[0, 0] - q)) / max(1.0, np.max(np.abs(q)))
        worst[kind] = max(worst[kind], err)
        if kind in (0, 2):
            assert np.array_equal(e.X[0, 0], q), (ci, d, temp, kind)     # elementwise arithmetic: bit-exact
    assert worst[1] < 1e-13                                               # AM: BLAS summation order differs


def test_welford_bit_exact(golden):
    g = golden("welford")
    for d in (5, 100):
        mu, M2 = np.zeros(d), np.zeros((d, d))
        for ep in range(3):
            tag = "d%d_e%d" % (d, ep)
            am = g["am_" + tag]
            cov = orc.welford(am, mu, M2, (ep + 1) * am.shape[0])
            assert np.array_equal(mu, g["mu_" + tag])
            assert np.array_equal(M2, g["M2_" + tag])
            assert np.array_equal(cov, g["cov_" + tag])


def test_de_buffer(golden):
    g = golden("debuffer")
    d, mem, burn = g["shape"]
    DE = np.zeros((burn, d))
    for ep in range(5):
        orc.de_update(DE, g["am_%d" % ep])
        assert np.array_equal(DE, g["de_%d" % ep])


def test_ptswap_decisions(golden):
    """PTswap root sweep (PTMCMCSampler.py:666-686): same uniforms -> same permutation and credits."""
    g = golden("ptswap")
    for ci, (n, d) in enumerate(g["meta"]):
        lnL, ladder, u = g["lnL_%d" % ci], g["ladder_%d" % ci], g["u_%d" % ci]
        m, acc = orc.swap_sweep(ladder, lnL, uniforms=u)
        assert np.array_equal(lnL[m[0]], g["newlnL_%d" % ci], equal_nan=True)
        assert np.array_equal(g["p0s_%d" % ci][m[0]], g["newp0s_%d" % ci])
        assert np.array_equal(acc[0].astype(float), g["acc_%d" % ci])
    # a state may travel several levels in one sweep (carried map)
    assert any(np.max(np.abs(orc.swap_sweep(g["ladder_%d" % c], g["lnL_%d" % c], uniforms=g["u_%d" % c])[0][0]
                             - np.arange(g["meta"][c][0]))) > 1 for c in range(len(g["meta"])))


def _engine_from_traj(g):
    d, n = int(g["ndim"]), int(g["nranks"])
    logl = ("dense", g["dense_mu"], g["dense_icov"]) if "dense_mu" in g else ("iso",)
    logp = ("box", g["box_lo"], g["box_hi"]) if "box_lo" in g else ("flat",)
    groups = None
    if "groups_flat" in g:
        groups = np.split(g["groups_flat"], np.cumsum(g["groups_size"])[:-1])
    e = orc.OracleEngine(d, n, 1, g["cov0"], ladder=g["ladder"], logl=logl, logp=logp, groups=groups,
                         weights=(int(g["kw_SCAMweight"]), int(g["kw_AMweight"]), int(g["kw_DEweight"])),
                         cov_update=int(g["kw_covUpdate"]), burn=int(g["kw_burn"]), tskip=int(g["kw_Tskip"]),
                         hot_chain=bool(g["hot"]))
    e.init_state(g["p0"])
    return e


@pytest.mark.parametrize("name", ["traj_single_d5", "traj_single_box_d4", "traj_pt4_d6", "traj_pt3_dense_d8",
                                  "traj_pt2_scam_d100", "traj_groups_d6", "traj_groups_pt2_d5"])
def test_full_trajectory_matches_reference(golden, name):
    """sample() end to end (PTMCMCSampler.py:495-629) replayed from each rank's recorded draws."""
    g = golden(name)
    e = _engine_from_traj(g)
    n, niter, thin = int(g["nranks"]), int(g["kw_Niter"]), int(g["kw_thin"])
    assert np.array_equal(e.temps_mh, g["temps"])
    replay = [(g["dk_%d" % r], g["dv_%d" % r], g["db_%d" % r]) for r in range(n)]
    epochs = []
    orig = e._svd

    def spy(w):
        orig(w)
        epochs.append((e.mu[0].copy(), e.M2[0].copy(), e.cov[0].copy(), None, e.S[0, 0, :e.gsize[0]].copy()))

    e._svd = spy
    rec = e.run(niter, replay=replay, record=True)
    assert e.replay_left == [0] * n                      # every recorded draw consumed, in kind and bound
    for r in range(n):
        ref_chain, ref_lnl, ref_lnp = g["chain_%d" % r], g["lnlike_%d" % r], g["lnprob_%d" % r]
        got = rec["X"][::thin, 0, r]
        assert got.shape == ref_chain.shape
        scale = max(1.0, np.max(np.abs(ref_chain)))
        assert np.max(np.abs(got - ref_chain)) / scale < 1e-11
        assert np.allclose(rec["lnL"][::thin, 0, r], ref_lnl, rtol=1e-10, atol=1e-10)
        assert np.allclose(rec["lnprob"][::thin, 0, r], ref_lnp, rtol=1e-10, atol=1e-10)
        # decisions are exact
        assert int(e.nacc[0, r]) == int(g["nacc_%d" % r])
        names = [str(s) for s in g["jnames_%d" % r]]
        for j, nm in enumerate(NAMES):
            want = g["jstats_%d" % r][names.index(nm)] if nm in names else [0, 0]
            assert list(e.jstat[0, r, j].astype(int)) == list(want), (r, nm)
        assert int(e.nswap[0, r]) == int(g["nswap_%d" % r])
        assert e.swap_proposed == int(g["swapprop_%d" % r])
    assert len(epochs) == int(g["nepochs"])
    for i, (mu, M2, cov, U, S) in enumerate(epochs):
        assert np.allclose(mu, g["ep_mu_%d" % i], rtol=1e-10, atol=1e-12)
        assert np.allclose(M2, g["ep_M2_%d" % i], rtol=1e-9, atol=1e-12)
        assert np.allclose(cov, g["ep_cov_%d" % i], rtol=1e-9, atol=1e-12)
        assert np.allclose(S, g["ep_S_%d" % i], rtol=1e-8, atol=1e-14)


def test_builtin_likelihood_gradients():
    """The analytic gradients the device NUTS / HMC use: the curved likelihood against the formulas of the reference's
    examples/curved_likelihood.ipynb (cell 1, lnlikefn / lnlikefn_grad, written out here), the Gaussians against their
    closed forms and all three against central differences."""
    import ctypes as C
    rng = np.random.default_rng(7)

    def oracle(kind, x, par=None):
        d = len(x)
        par_l = np.zeros(1) if par is None else par
        cfg = orc.Cfg(ndim=d, ntemps=1, nwalkers=1, lanes=4, logl_kind=orc.LOGL[kind], logp_kind=0, logl_par=orc._p(par_l))
        g = np.zeros(d)
        v = orc.lib().orc_logl_grad(C.byref(cfg), orc._p(np.ascontiguousarray(x)), orc._p(g))
        return v, g

    def curved_nb(x):                                   # one 2-d block of the notebook's likelihood
        l0 = -x[0] ** 2 - (9 + 4 * x[0] ** 2 + 9 * x[1]) ** 2
        l1 = -8 * x[0] ** 2 - 8 * (x[1] - 2) ** 2
        g0 = np.array([-2.0 * x[0] - 2.0 * (9 + 4 * x[0] ** 2 + 9 * x[1]) * (8 * x[0]), -18.0 * (9 + 4 * x[0] ** 2 + 9 * x[1])])
        g1 = np.array([-16 * x[0], -16 * (x[1] - 2)])
        lik = np.exp(l0) + 0.5 * np.exp(l1)
        return np.log(lik), (np.exp(l0) * g0 + 0.5 * np.exp(l1) * g1) / lik

    for _ in range(20):
        x = np.concatenate([np.array([-0.1, -0.5]) + rng.normal(size=2) * 0.3 for _ in range(3)])
        v, g = oracle("curved", x)
        ref = [curved_nb(x[i:i + 2]) for i in range(0, 6, 2)]
        assert abs(v - sum(r[0] for r in ref)) < 1e-12 * max(1.0, abs(v))
        np.testing.assert_allclose(g, np.concatenate([r[1] for r in ref]), rtol=1e-11, atol=1e-12)
    d = 6
    A = rng.normal(size=(d, d))
    P, mu = A @ A.T / d + np.eye(d), rng.normal(size=d)
    par = orc.dense_par(mu, P)
    for kind, p, f in (("iso", None, lambda x: -0.5 * x @ x), ("dense", par, lambda x: -0.5 * (x - mu) @ P @ (x - mu))):
        x = rng.normal(size=d)
        v, g = oracle(kind, x, p)
        assert abs(v - f(x)) < 1e-12 * max(1.0, abs(v))
        num = np.array([(f(x + h) - f(x - h)) / 2e-6 for h in np.eye(d) * 1e-6])
        np.testing.assert_allclose(g, num, rtol=1e-6, atol=1e-6)


def test_interval_family_values_and_gradients_are_the_reference_workloads(golden):
    """LOGL_INTERVAL (include/ptmi.h PTMI_LOGL_INTERVAL) against the reference's own test likelihood -- tests/test_nuts.py
    GaussianLikelihood inside intervalTransform, evaluated by the reference's classes themselves (make_golden.py gen_interval): the
    40-d box (0, 10) of test_nuts and an uneven 7-d box; in every lane count the kernels use.  Tolerance: the oracle's own exp / log
    (<= 1 ulp each) and a different summation order, 1e-12 relative."""
    import ctypes as C
    g = golden("interval")
    for tag in ("t40", "u7"):
        a, b, P = g[tag + "_a"], g[tag + "_b"], g[tag + "_p"]
        d = len(a)
        par = orc.interval_par(a, b, d)
        assert np.all(g[tag + "_lp"] == 0.0)            # the workload's prior (the box in x) is 0 at every finite p: ("flat",)
        for lanes in (4, 16, 64):
            cfg = orc.Cfg(ndim=d, ntemps=1, nwalkers=1, lanes=lanes, logl_kind=orc.LOGL["interval"], logp_kind=0, logl_par=orc._p(par))
            for k, p in enumerate(P):
                gr = np.zeros(d)
                v = orc.lib().orc_logl_grad(C.byref(cfg), orc._p(np.ascontiguousarray(p)), orc._p(gr))
                assert abs(v - g[tag + "_ll"][k]) <= 1e-12 * abs(g[tag + "_ll"][k]), (tag, lanes, k)
                np.testing.assert_allclose(gr, g[tag + "_grad"][k], rtol=1e-11, atol=1e-13)
                assert v == orc.lib().orc_logl(C.byref(cfg), orc._p(np.ascontiguousarray(p)))
    # overflow of e^p: inf / inf, as in the reference (its prior turns the NaN into -inf; here the NaN fails every comparison of the accept step)
    cfg = orc.Cfg(ndim=2, ntemps=1, nwalkers=1, lanes=4, logl_kind=orc.LOGL["interval"], logp_kind=0, logl_par=orc._p(orc.interval_par(0.0, 10.0, 2)))
    assert np.isnan(orc.lib().orc_logl(C.byref(cfg), orc._p(np.array([800.0, 0.0]))))
    assert np.isfinite(orc.lib().orc_logl(C.byref(cfg), orc._p(np.array([-800.0, 0.0]))))
