"""HIP path (through the C ABI) against the CPU oracle, same Philox seeds: bit-exact.

Run on the GPU box: ``python -m pytest tests -m gpu``.  Nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from oracle import oracle as orc
    from ptmcmcsampler_amd import _lib
    from ptmcmcsampler_amd.engine import PTEngine
    _lib.load()
    assert _lib.device_count() >= 1, "no MI355X visible"
    return orc, _lib, PTEngine


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_same(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    if a.dtype.kind == "f":
        bad = _bits(a) != _bits(b)
        # all NaNs are one value for this purpose
        bad &= ~(np.isnan(a) & np.isnan(b))
        assert not bad.any(), "%s: %d of %d differ, first at %s: %r vs %r" % (
            what, bad.sum(), bad.size, np.argwhere(bad)[0], a[tuple(np.argwhere(bad)[0])], b[tuple(np.argwhere(bad)[0])])
    else:
        assert np.array_equal(a.astype(np.int64), b.astype(np.int64)), what


def test_device_math_is_bit_identical_to_oracle(mods):
    orc, _lib, _ = mods
    L, O = _lib.load(), orc.lib()
    rs = np.random.RandomState(3)

    def dev(op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = x if y is None else np.ascontiguousarray(y, dtype=np.float64)
        out = np.empty_like(x)
        _lib.check(L.ptmi_selftest_math(0, op, x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size))
        return out

    xs = np.concatenate([rs.rand(20000), 2.0 ** rs.uniform(-1000, 1000, 5000), [0.0, 1.0, 2.0 ** -53, np.inf, 5e-324, 3e-310]])
    assert_same(dev(0, xs), [O.orc_log(float(v)) for v in xs], "log")
    es = np.concatenate([rs.uniform(-750, 710, 20000), rs.uniform(-3, 3, 5000), [0.0, -745.2, 709.9, -np.inf, np.inf, np.nan]])
    assert_same(dev(1, es), [O.orc_exp(float(v)) for v in es], "exp")
    us = np.concatenate([rs.rand(20000), [0.0, 0.25, 0.5, 0.75, 1 - 2.0 ** -53]])
    assert_same(dev(2, us), [O.orc_cos2pi(float(v)) for v in us], "cos2pi")
    ps = np.concatenate([rs.rand(10000) * 1e3, 2.0 ** rs.uniform(-1000, 1000, 5000)])
    assert_same(dev(3, ps), np.sqrt(ps), "sqrt is correctly rounded")
    a, b = rs.randn(20000) * 10 ** rs.uniform(-5, 5, 20000), rs.randn(20000) * 10 ** rs.uniform(-5, 5, 20000)
    assert_same(dev(4, a, b), a / b, "division is correctly rounded")
    w = rs.randint(0, 2 ** 62, size=(5000, 2)).astype(np.uint64) * np.uint64(4) + np.uint64(3)
    got = dev(5, w[:, 0].view(np.float64), w[:, 1].view(np.float64))
    assert_same(got, [O.orc_normal(int(p), int(q)) for p, q in w], "normal")
    # the table-driven draws of the MH path: ln of the (0,1] uniform of a word, cos / sin of the angle of a word, the SCAM normal
    import ctypes as C
    ww = np.concatenate([w.reshape(-1), np.array([0, 2 ** 64 - 1, 2 ** 11, (2 ** 53 - 1) << 11, (2 ** 53 - 2) << 11, 2 ** 59 - 1, 2 ** 59, 2 ** 58], dtype=np.uint64),
                         (np.uint64(2 ** 53 - 1) - np.arange(1, 3000, dtype=np.uint64)) << np.uint64(11)])
    assert_same(dev(10, ww.view(np.float64)), [O.orc_unit_log(int(v)) for v in ww], "unit_log")
    sn, cs = C.c_double(), C.c_double()
    want = []
    for v in ww:
        O.orc_unit_sincos64(int(v), C.byref(sn), C.byref(cs))
        want.append((cs.value, sn.value))
    want = np.array(want)
    assert_same(dev(11, ww.view(np.float64)), want[:, 0], "unit cos")
    assert_same(dev(12, ww.view(np.float64)), want[:, 1], "unit sin")
    hh = (ww[:len(w)] & np.uint64(0xFFFFFFFF))
    assert_same(dev(13, hh.view(np.float64), ww[::-1][:len(w)].copy().view(np.float64)),
                [O.orc_unit_normal32(int(q), int(h)) for h, q in zip(hh, ww[::-1][:len(w)])], "SCAM normal")
    # ... and their distribution at a size the CPU tests do not reach: 4 M device normals of either kind, moments within
    # five standard errors and a chi-square over 256 equiprobable bins
    from scipy import stats
    nbig = 1 << 22
    big = rs.randint(0, 2 ** 63, size=(nbig, 2), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(nbig, 2)).astype(np.uint64)
    z_scam = dev(13, (big[:, 1] & np.uint64(0xFFFFFFFF)).view(np.float64), big[:, 0].copy().view(np.float64))
    rad = np.sqrt(-2.0 * dev(10, big[:, 0].copy().view(np.float64)))
    z_cos, z_sin = rad * dev(11, big[:, 1].copy().view(np.float64)), rad * dev(12, big[:, 1].copy().view(np.float64))
    edges = stats.norm.ppf(np.arange(1, 256) / 256.0)
    for name, z in (("SCAM", z_scam), ("AM cos", z_cos), ("AM sin", z_sin)):
        assert np.isfinite(z).all(), name
        se = 1.0 / np.sqrt(nbig)
        assert abs(z.mean()) < 5 * se and abs(z.var() - 1) < 5 * np.sqrt(2.0) * se, name
        assert abs((z ** 3).mean()) < 5 * np.sqrt(15.0) * se and abs((z ** 4).mean() - 3) < 5 * np.sqrt(96.0) * se, name
        counts = np.bincount(np.searchsorted(edges, z), minlength=256)
        chi2 = ((counts - nbig / 256.0) ** 2 / (nbig / 256.0)).sum()
        assert stats.chi2.sf(chi2, 255) > 1e-4, (name, chi2)
    assert abs(np.corrcoef(z_cos, z_sin)[0, 1]) < 5.0 / np.sqrt(nbig)
    # the 16-lane butterfly: every lane of a row gets the oracle's tree sum
    v = rs.randn(64 * 8)
    got = dev(6, v)
    for r in range(len(v) // 16):
        p = v[r * 16:(r + 1) * 16].copy()
        m = 8
        while m >= 1:
            p = p + p[np.arange(16) ^ m]
            m >>= 1
        assert_same(got[r * 16:(r + 1) * 16], p, "group_sum")


def test_device_philox_matches_oracle(mods):
    orc, _lib, _ = mods
    L = _lib.load()
    rs = np.random.RandomState(4)
    ck = rs.randint(0, 2 ** 32, size=(4096, 6), dtype=np.uint64).astype(np.uint32)
    ck[0] = 0
    ck[1] = 0xFFFFFFFF
    out = np.zeros((len(ck), 4), dtype=np.uint32)
    _lib.check(L.ptmi_selftest_philox(0, ck.ctypes.data, out.ctypes.data, len(ck)))
    for i in range(0, len(ck), 37):
        assert list(out[i]) == orc.philox([int(v) for v in ck[i, :4]], [int(ck[i, 4]), int(ck[i, 5])])
    assert list(out[0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]


def _pair(mods, d, nt, W, **kw):
    orc, _lib, PTEngine = mods
    rs = np.random.RandomState(kw.pop("rs", 0))
    cov0 = kw.pop("cov0", None)
    if cov0 is None:
        A = rs.randn(d, d)
        cov0 = (A @ A.T / d + 0.5 * np.eye(d)) * 0.01
    p0 = kw.pop("p0", None)
    if p0 is None:
        p0 = rs.randn(W, nt, d) * 0.3
    g = PTEngine(d, nt, W, cov0, **kw)
    okw = {k: v for k, v in kw.items() if k not in ("split", "use_de_buffer", "stats_async")}    # (stats_async: scheduling only, the oracle has no such thing)
    o = orc.OracleEngine(d, nt, W, cov0, **okw)
    assert o.lanes == _lib.lanes_for(d)
    g.init_state(p0)
    o.init_state(p0)
    return g, o


def _compare(g, o, what=""):
    g.sync()
    for name in ("X", "lnL", "lp", "temp_of", "slot_of", "nacc", "jstat"):
        assert_same(g.get(name), getattr(o, name), what + name)
    assert_same(g.get("nswap"), o.nswap, what + "nswap")
    if g.owns_cold and getattr(g, "am_rle", False):
        lo, hi = g.am_period()                                   # am_mode "rle": the ring keeps the rows of the current covariance period
        rows = np.arange(lo, hi + 1) % g.cov_update
        assert_same(g.get("AM")[:, rows], o.AM[:, rows], what + "AM (current period)")
    elif g.owns_cold:
        assert_same(g.get("AM"), o.AM, what + "AM")


def test_initial_evaluation(mods):
    for d, logl in ((100, ("iso",)), (20, ("iso",)), (1000, ("iso",)), (6, ("curved",))):
        g, o = _pair(mods, d, 3, 5, logl=logl, tskip=0)
        _compare(g, o, "init d=%d " % d)


def test_scam_iso_fused_steps_bit_exact(mods):
    """The bench slice: SCAM + iso-Gaussian + accept, K fused steps (config 2 shape, small batch)."""
    g, o = _pair(mods, 100, 8, 16, weights=(20, 0, 0), tskip=0, cov_update=64, seed=1234)
    g.mh_steps(1, 40)
    o.run(40)
    _compare(g, o, "scam ")
    assert o.nacc.sum() > 0 and o.jstat[..., 0, 0].sum() == 8 * 16 * 40


@pytest.mark.parametrize("d,nt,W", [(5, 4, 6), (20, 3, 7), (100, 4, 5), (300, 2, 3), (1000, 2, 2)])
def test_full_cycle_with_adaptation_and_swaps(mods, d, nt, W):
    """SCAM+AM+DE, covariance epochs, DE epochs and activation, PT swaps: every array bit-exact."""
    n = 330 if d <= 100 else 130
    g, o = _pair(mods, d, nt, W, weights=(20, 20, 20), cov_update=50, burn=100, tskip=10, seed=77, rs=d)
    g.run(n)
    o.run(n)
    _compare(g, o, "full d=%d " % d)
    assert_same(g.get("mu"), o.mu, "mu")
    assert_same(g.get("M2"), o.M2, "M2")
    assert_same(g.get("cov"), o.cov, "cov")
    assert_same(g.get("Ut"), o.Ut, "Ut")
    assert g.swap_proposed == o.swap_proposed and o.nswap.sum() > 0
    if n > 200:
        assert o.jstat[..., 2, 0].sum() > 0     # DE was proposed after burn


def test_dense_gaussian_box_prior_hot_chain(mods):
    d = 8
    rs = np.random.RandomState(5)
    mu = rs.uniform(0, 10, d)
    A = rs.randn(d, d)
    P = np.linalg.inv(A @ A.T + np.eye(d))
    g, o = _pair(mods, d, 3, 6, logl=("dense", mu, P), logp=("box", np.zeros(d), 10 * np.ones(d)),
                 p0=rs.uniform(0, 10, (6, 3, d)), cov0=np.eye(d) * 0.5, weights=(20, 20, 20), cov_update=40,
                 burn=80, tskip=8, hot_chain=True, seed=5)
    g.run(250)
    o.run(250)
    _compare(g, o, "dense ")
    assert (o.jstat[..., 0].sum(-1) > o.jstat[..., 1].sum(-1)).all()      # prior rejections happened


def test_dense_d100(mods):
    d = 100
    rs = np.random.RandomState(0)
    A = rs.standard_normal((d, d))
    P = np.linalg.inv(A @ A.T / d + np.eye(d))
    g, o = _pair(mods, d, 4, 3, logl=("dense", np.zeros(d), P), cov0=np.eye(d) * 0.01, weights=(20, 20, 0),
                 cov_update=30, burn=1000, tskip=10, seed=9)
    g.run(70)
    o.run(70)
    _compare(g, o, "dense100 ")


def test_curved_likelihood(mods):
    g, o = _pair(mods, 20, 4, 5, logl=("curved",), logp=("box", -10 * np.ones(20), 10 * np.ones(20)),
                 cov0=np.eye(20), weights=(10, 0, 10), cov_update=50, burn=100, tskip=10, seed=3)
    g.run(260)
    o.run(260)
    _compare(g, o, "curved ")


@pytest.mark.parametrize("d,W,n", [(12, 9, 250), (100, 70, 130), (300, 5, 130)])
def test_pooled_covariance_mode(mods, d, W, n):
    """One covariance from all walkers: fused Welford (matrix cores up to d = 112, tiles beyond), two-level pooling,
    symmetric eigen-decomposition on the host (8 BLAS threads beyond 256 parameters)."""
    g, o = _pair(mods, d, 3, W, cov_mode="pooled", weights=(20, 20, 20), cov_update=40, burn=80, tskip=10, seed=21)
    g.run(n)
    o.run(n)
    _compare(g, o, "pooled ")
    assert_same(g.get("cov"), o.cov, "pooled cov")
    assert_same(g.get("Ut"), o.Ut, "pooled Ut")


def test_welford_kernel_against_reference_fixture(mods, golden):
    """_updateRecursive (PTMCMCSampler.py:769-794): the device kernel reproduces the reference's mu, M2, cov bit for bit."""
    orc, _lib, PTEngine = mods
    gld = golden("welford")
    for d in (5, 100):
        mem = gld["am_d%d_e0" % d].shape[0]
        g = PTEngine(d, 1, 2, np.eye(d), weights=(1, 0, 0), cov_update=mem, burn=mem, tskip=0)
        for ep in range(3):
            tag = "d%d_e%d" % (d, ep)
            am = gld["am_" + tag]
            g.put("AM", np.stack([am, am[::-1]]))        # walker 1 sees the rows reversed: must differ
            _lib.check(g.lib.ptmi_update_cov(g.h, (ep + 1) * mem))
            assert_same(g.get("mu")[0], gld["mu_" + tag], "mu")
            assert_same(g.get("M2")[0], gld["M2_" + tag], "M2")
            assert_same(g.get("cov")[0], gld["cov_" + tag], "cov")
        assert not np.array_equal(g.get("M2")[1], g.get("M2")[0])


def test_de_ring_against_reference_fixture(mods, golden):
    """_updateDEbuffer (PTMCMCSampler.py:806-817) as a ring: logical row r == reference row r."""
    orc, _lib, PTEngine = mods
    gld = golden("debuffer")
    d, mem, burn = [int(v) for v in gld["shape"]]
    g = PTEngine(d, 1, 1, np.eye(d), weights=(1, 0, 1), cov_update=mem, burn=burn, tskip=0)
    head = 0
    for ep in range(5):
        g.put("AM", gld["am_%d" % ep][None])
        g.update_de()
        head = (head + mem) % burn
        ring = g.get("DE")[0]
        logical = np.roll(ring, -head, axis=0)
        assert_same(logical, gld["de_%d" % ep], "DE epoch %d" % ep)


def test_split_path_equals_fused(mods):
    """propose -> (host evaluates logl/logp) -> accept gives the fused kernel's result."""
    orc, _lib, PTEngine = mods
    import torch
    d, nt, W = 10, 3, 4
    g, o = _pair(mods, d, nt, W, weights=(20, 20, 0), cov_update=500, burn=1000, tskip=0, seed=8, split=True)
    for it in range(1, 31):
        _lib.check(g.lib.ptmi_propose(g.h, it))
        Q = g.t["Q"]
        newl = (-0.5 * (Q * Q).sum(-1))
        # the host's summation order differs in the last bits; take the oracle's value so decisions match
        qn = Q.cpu().numpy()
        newl = torch.from_numpy(np.array([[orc.lib().orc_logl(C.byref(o.cfg), qn[w, s].ctypes.data_as(orc._dp))
                                           for s in range(nt)] for w in range(W)])).to(g.device)
        newp = torch.zeros_like(newl)
        _lib.check(g.lib.ptmi_accept(g.h, it, newl.data_ptr(), newp.data_ptr()))
    o.run(30)
    _compare(g, o, "split ")


def test_full_size_properties_and_subset_parity(mods):
    """BASELINE config 2 shape (64 temps x 4096 walkers x 100-d): invariants on the whole batch and
    bit parity on a few walkers (walkers are independent and the RNG is keyed by global walker id)."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    g = PTEngine(d, nt, W, cov0, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=1234,
                 cov_mode="pooled", use_de_buffer=False)
    g.init_state(np.zeros(d))
    n = 200
    g.run(n)
    g.sync()
    flags, G, E = g.last_variant()
    # what bench.py times: persistent blocks over one LDS copy of the pooled table, exact shape
    assert flags & _lib.VAR_PERSISTENT and flags & _lib.VAR_LDS_UT and flags & _lib.VAR_LDS_DRAWT and (G, E) == (4, 25)
    so, to = g.get("slot_of"), g.get("temp_of")
    assert (np.sort(so, axis=1) == np.arange(nt)).all()                       # tables stay permutations
    assert (np.take_along_axis(to, so.astype(np.int64), 1) == np.arange(nt)).all()
    js = g.get("jstat").astype(np.int64)
    assert (js[..., 0].sum(-1) == n).all()                                     # every chain proposed n times
    assert (js[..., 1].sum(-1) == g.get("nacc").astype(np.int64)).all()
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * (X ** 2).sum(-1), rtol=1e-12)
    acc = g.get("nacc").astype(float).mean(0) / n
    assert 0.5 < acc.min() and acc.max() <= 1.0 and acc[0] < acc[-1]        # small cov0: high, and hotter accepts more
    assert g.get("nswap").sum() > 0
    # idempotence: re-evaluating the state changes nothing
    _lib.check(g.lib.ptmi_eval_state(g.h))
    assert_same(g.get("lnL"), lnL, "eval_state idempotent")
    for w0 in (0, 1777, 4095):
        o = orc.OracleEngine(d, nt, 1, cov0, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=1234,
                             cov_mode="pooled", walker0=w0)
        o.init_state(np.zeros(d))
        o.run(n)
        assert_same(X[w0], o.X[0], "walker %d X" % w0)
        assert_same(so[w0], o.slot_of[0], "walker %d slot_of" % w0)
        assert_same(g.get("nswap")[w0], o.nswap[0], "walker %d nswap" % w0)


@pytest.mark.parametrize("d,nt,W", [(1, 1, 1), (2, 1, 3), (3, 2, 1), (7, 5, 13), (33, 3, 11), (97, 3, 5), (99, 2, 7), (101, 3, 5), (104, 2, 9), (105, 2, 5), (417, 2, 2), (641, 2, 2), (1025, 2, 2),
                                    (3, 300, 5), (2, 520, 37)])     # long ladders: the swap sweep stages fewer walkers per block
def test_ragged_and_boundary_sizes(mods, d, nt, W):
    """Sizes that do not fill a block or a shape: one dimension, one temperature (no swaps), chain counts that
    are not multiples of the block, the last ndim of the 4-lane shapes (104) and the first of the next ones."""
    kw = dict(weights=(20, 20, 20), cov_update=30, burn=60, tskip=7 if nt > 1 else 0, seed=d * 100 + nt, rs=d + 1)
    g, o = _pair(mods, d, nt, W, **kw)
    n = 150 if d < 200 else 70
    g.run(n)
    o.run(n)
    _compare(g, o, "ragged d=%d nt=%d W=%d " % (d, nt, W))
    assert_same(g.get("cov"), o.cov, "cov")


@pytest.mark.parametrize("prior", ["flat", "box"])
@pytest.mark.parametrize("d,nt,W,cu,tskip", [(105, 3, 5, 30, 7), (130, 1, 7, 50, 0), (208, 2, 3, 50, 45), (300, 3, 4, 40, 13), (416, 2, 3, 30, 7), (417, 2, 3, 70, 0),
                                             (512, 1, 2, 70, 0), (641, 3, 2, 50, 45), (1000, 4, 3, 70, 33), (1024, 2, 2, 30, 7), (1025, 2, 2, 40, 39)])
def test_wide_shapes_scam_only_on_the_padded_table(mods, d, nt, W, cu, tskip, prior, monkeypatch):
    """16 and 64 lanes per chain (ndim > 104), SCAM-only cycle, ONE table for the launch (pooled covariance): the wide draw batches
    (all lanes of a chain draw: 8 / 32 iterations per pass; PTMCMCSampler.py:820-876) and the direction read from the library's
    zero-padded copy of the table (include/ptmi.h PTMI_VAR_UTPAD).  Launch lengths that are no multiple of a pass, launches longer
    than one (tskip = 0: a launch per covariance period), covariance epochs between the launches (the copy follows the table); the
    same run with the copy switched off (PTMI_NO_UTPAD is read once per process, so that variant is checked through per-walker tables)."""
    orc, _lib, PTEngine = mods
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=tskip, seed=d * 10 + nt, rs=d + 2, cov_mode="pooled")
    if prior == "box":
        rs = np.random.RandomState(d)
        kw.update(logp=("box", -0.6 - rs.rand(d) * 0.1, 0.6 + rs.rand(d) * 0.1), p0=rs.uniform(-0.3, 0.3, (W, nt, d)))
    g, o = _pair(mods, d, nt, W, **kw)
    for n in (cu + 3, 1, 37, 2 * cu):
        g.run(n)
        o.run(n)
        _compare(g, o, "wide d=%d it=%d " % (d, g.iter))
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_UTPAD and G == (16 if d <= 416 else 64)
    assert_same(g.get("cov"), o.cov, "cov")
    assert_same(g.get("Ut"), o.Ut, "Ut")


@pytest.mark.parametrize("d,cov_mode,weights", [(130, "per_walker", (20, 0, 0)), (641, "per_walker", (20, 0, 0)), (300, "pooled", (20, 20, 20)),
                                                (1000, "pooled", (20, 0, 20)), (417, "per_walker", (20, 20, 20))])
def test_wide_draw_batches_in_the_other_wide_kernels(mods, d, cov_mode, weights):
    """The wide draw batches in the kernels that do not read the padded copy: a table per walker, and cycles with AM / DE entries
    (DrawBatch: P and Q words of 8 / 32 iterations per pass), walker picks included."""
    orc, _lib, PTEngine = mods
    for pick in ("chain", "walker"):
        kw = dict(weights=weights, cov_update=40, burn=80, tskip=11, seed=d + 5, rs=d, cov_mode=cov_mode, pick_mode=pick)
        g, o = _pair(mods, d, 3, 4, **kw)
        for n in (43, 90, 57):
            g.run(n)
            o.run(n)
            _compare(g, o, "wide-other d=%d %s it=%d " % (d, pick, g.iter))
        flags, G, E = g.last_variant()
        assert G in (16, 64) and (sum(weights[1:]) > 0) == bool(flags & _lib.VAR_FULL)
        if sum(weights[1:]) == 0:
            assert not flags & _lib.VAR_UTPAD


@pytest.mark.parametrize("pers", [512, 0])
@pytest.mark.parametrize("d,prior", [(99, "flat"), (100, "flat"), (100, "box"), (101, "flat"), (104, "box"), (81, "flat")])
def test_scam_only_table_kernel_around_the_exact_shape(mods, d, prior, pers, monkeypatch):
    """SCAM-only cycle with the eigenvector table in LDS (the config-2 bench kernel) on both sides of ndim = 100: 100 runs the
    EXACT shape (4, 25) -- no bounds checks, table rows stored in lane order and read 16 bytes at a time, chain-scalar half of
    the proposal computed in the draw pass -- its neighbours the general shape (4, 26).  Pooled covariance epochs in between
    change the table; 70 walkers x 4 ranks fill more than one block (PTMCMCSampler.py:820-876, 605-622).
    ``pers``: with one table for the whole launch (pooled covariance) the launch goes to PERSISTENT blocks of 512 threads,
    one per CU over one LDS copy of the table, every wave walking over units of 16 chains (the default and what bench.py
    times); 0 = the kernel with a table copy per block of 64 chains, which per-walker tables keep."""
    orc, _lib, _ = mods
    monkeypatch.setenv("PTMI_ULDS_PERS", str(pers))
    kw = dict(weights=(20, 0, 0), cov_update=40, burn=1000, tskip=10, seed=d, cov_mode="pooled", cov0=np.eye(d) * 0.02)
    if prior == "box":
        kw.update(logp=("box", -0.4 * np.ones(d), 0.5 * np.ones(d)), p0=np.random.RandomState(d).uniform(-0.1, 0.1, (70, 4, d)))
    g, o = _pair(mods, d, 4, 70, **kw)
    for n in (97, 3, 50):                                      # odd launch lengths: the draw pass serves two steps
        g.run(n)
        o.run(n)
    flags, G, E = g.last_variant()
    assert not flags & _lib.VAR_STAGED and not flags & _lib.VAR_FULL
    assert bool(flags & _lib.VAR_PERSISTENT) == (pers != 0)
    # a copy per block: two blocks' tables (+ sqrt(S) or the bounds) fit the CU up to ndim = 100; one copy per CU: every 4-lane shape
    assert bool(flags & _lib.VAR_LDS_UT) == (d <= 100 or pers != 0)
    assert (G, E) == ((4, 25) if d == 100 else (4, 26) if d > 80 else (4, 20))
    _compare(g, o, "scam table d=%d %s pers=%d " % (d, prior, pers))
    assert_same(g.get("Ut"), o.Ut, "Ut")


@pytest.mark.parametrize("fusedsweep", [1, 0])
@pytest.mark.parametrize("nt,W", [(2, 70), (3, 5), (9, 131), (10, 64), (17, 33), (64, 100), (130, 65), (257, 40), (300, 9), (520, 37)])
def test_swap_sweep_with_records_made_in_the_block(mods, nt, W, fusedsweep, monkeypatch):
    """PTswap (PTMCMCSampler.py:631-697) by swap_fused_kernel -- the records of a batch of eight pairs made by the block's other
    waves while wave 0 runs the hot -> cold recurrence on the previous batch -- and by the two-kernel form (PTMI_SWAP_FUSED=0:
    swap_prepare_kernel + swap_sweep_kernel, which very long ladders keep): ladders whose pair count is and is not a multiple
    of the batch, fewer pairs than one batch, walkers that do and do not fill the block (64 / 32 / 16 walkers per block as the
    ladder grows), the swap iteration's AM row stored by the write-out.  Every table, counter and AM row against the oracle."""
    orc, _lib, _ = mods
    monkeypatch.setenv("PTMI_SWAP_FUSED", str(fusedsweep))
    d = 3
    g, o = _pair(mods, d, nt, W, weights=(20, 20, 0), cov_update=16, burn=1000, tskip=3, seed=17 * nt + W)
    for n in (7, 12, 17):
        g.run(n)
        o.run(n)
        _compare(g, o, "sweep nt=%d W=%d fused=%d " % (nt, W, fusedsweep))
    assert o.nswap.sum() > 0 and g.swap_proposed == 12


@pytest.mark.parametrize("d,nt,W", [(100, 16, 37), (100, 32, 70), (100, 48, 19), (64, 64, 9), (100, 64, 130)])
def test_persistent_kernel_cold_first_walk(mods, d, nt, W):
    """The persistent SCAM kernel walks its units of 16 chains cold-first when a walker's ranks fill whole units (ntemps a
    multiple of 16): position p < W of the order is walker p's cold unit (``slot_of[p][0] >> 4``: it moves with every swap), the
    others follow walker by walker.  One, two, three and four units per walker, more units than the 2048 waves of a launch
    (130 walkers x 4) and fewer; swaps every 7 iterations move the cold chain between the units; the AM ring (written by the
    cold units and by the sweep's write-out at swap iterations) and the pooled covariance it feeds are compared bit for bit
    (PTMCMCSampler.py:327-328, 624-627, 820-876)."""
    orc, _lib, _ = mods
    g, o = _pair(mods, d, nt, W, weights=(20, 0, 0), cov_update=30, burn=1000, tskip=7, seed=5 * d + nt, cov_mode="pooled",
                 cov0=np.eye(d) * 0.02)
    for n in (29, 40, 31):
        g.run(n)
        o.run(n)
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_PERSISTENT and flags & _lib.VAR_LDS_UT
        _compare(g, o, "cold first d=%d nt=%d W=%d " % (d, nt, W))
    assert_same(g.get("cov"), o.cov, "cov")
    assert o.nswap[:, 0].sum() > 0                              # the cold chain did change places


@pytest.mark.parametrize("d,nt,W,pick", [(130, 3, 7, "chain"), (300, 2, 5, "chain"), (417, 3, 3, "walker"), (640, 2, 3, "chain"), (1000, 3, 2, "chain")])
def test_am_increments_ahead_of_the_launch_large_ndim(mods, d, nt, W, pick):
    """ndim > 104 with one pooled table: the AM increments U (cd sqrt(S) z) (PTMCMCSampler.py:879-933) of a piece of the launch
    are computed ahead of it by am_gemm_kernel on the matrix cores (16- and 64-lane shapes; one and two blocks per event
    tile), the step kernel reads them.  Same weights, same k-ascending fma chain as the oracle's: bit for bit, through
    covariance epochs that change the table, DE activation, launches of odd lengths and a scratch budget that cuts the
    launches into pieces of a few steps."""
    orc, _lib, _ = mods
    g, o = _pair(mods, d, nt, W, weights=(20, 20, 20), cov_update=30, burn=60, tskip=7, seed=d, cov_mode="pooled", pick_mode=pick,
                 cov0=np.eye(d) * 0.01)
    for n in (33, 5, 52):
        g.run(n)
        o.run(n)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_FULL and G == (16 if d <= 416 else 64)
    _compare(g, o, "AM ahead d=%d " % d)
    assert_same(g.get("Ut"), o.Ut, "Ut")
    assert o.jstat[..., 1, 0].sum() > 0 and o.jstat[..., 1, 1].sum() > 0 and o.jstat[..., 2, 0].sum() > 0


def test_minus_inf_start_and_nan_safety(mods):
    """A start outside the prior (lnL = lp = -inf, PTMCMCSampler.py:481-483) can only leave through a finite proposal."""
    d = 4
    p0 = np.zeros((3, 2, d))
    p0[0] = 5.0                       # walker 0 starts outside the box
    g, o = _pair(mods, d, 2, 3, logp=("box", -np.ones(d), np.ones(d)), p0=p0, cov0=np.eye(d) * 4.0,
                 weights=(20, 20, 0), cov_update=50, burn=100, tskip=5, seed=4)
    assert np.isneginf(o.lnL[0]).all() and np.isneginf(g.get("lnL")[0]).all()
    g.run(120)
    o.run(120)
    _compare(g, o, "inf-start ")


@pytest.mark.parametrize("d,groups", [(6, [[0, 1, 2], [3, 4, 5]]), (5, [[0, 1, 2, 3, 4], [3, 1], [2]]), (40, [list(range(0, 40, 2)), list(range(1, 40, 2)), [7]])])
def test_parameter_groups(mods, d, groups):
    """Per-group SVD and group-restricted SCAM / AM / DE (PTMCMCSampler.py:129-145, 839, 897, 955)."""
    g, o = _pair(mods, d, 3, 5, groups=groups, weights=(20, 20, 20), cov_update=40, burn=80, tskip=10, seed=13, rs=d)
    g.run(250)
    o.run(250)
    _compare(g, o, "groups d=%d " % d)
    assert_same(g.get("Ut"), o.Ut, "Ut")
    assert_same(g.get("S"), o.S, "S")
    assert o.jstat[..., :3, 0].sum(axis=(0, 1)).min() > 0


@pytest.mark.parametrize("d,groups", [(6, [[0, 1, 2], [3, 4, 5]]), (5, [[0, 1, 2, 3, 4], [3, 1], [2]]), (40, [list(range(0, 40, 2)), list(range(1, 40, 2)), [7]]),
                                      (100, [list(range(100)), list(range(0, 50)), list(range(50, 100)), [3, 99, 41]]),
                                      (130, [list(range(130)), list(range(5, 70))]), (500, [list(range(500)), list(range(100, 140))])])
@pytest.mark.parametrize("pieces", [False, True])
@pytest.mark.parametrize("cov_mode", ["pooled", "per_walker"])
def test_parameter_groups_with_am_increments_ahead_of_the_launch(mods, d, groups, pieces, cov_mode, monkeypatch):
    """Parameter groups with ONE pooled covariance (PTMCMCSampler.py:129-145, 897-933): a chain's AM pick has its own group, hence its
    own table, so the step kernels' matrix-core product (one table for the 16 chains of a wave) does not apply; the increments
    U_g (cd sqrt(S_g) z) of a piece of the launch are computed ahead of it by am_gemm_kernel, one launch per group over the listed
    picks (round 5; every shape, 4 / 16 / 64 lanes per chain) -- the same k-ascending fma chain over the group's rows as the step
    kernel's own vector-pipe product, which the same cases run with PTMI_NO_AM_AHEAD through the oracle.  `pieces`: a scratch of
    1 MB cuts the launches into single steps.  cov_mode "per_walker" (what real PTA runs combine: groups and a covariance per
    replica): an event's table is its walker's group table, the lists are per (walker, group), a grid row of the product per walker."""
    if pieces:
        monkeypatch.setenv("PTMI_AM_BUDGET_MB", "1")
    nt, W = 3, 5
    if cov_mode == "per_walker" and d > 130:
        pytest.skip("per-walker covariances at 500-d: the oracle's 5 x 500 x 500 SVDs per epoch (covered at 130-d)")
    g, o = _pair(mods, d, nt, W, groups=groups, weights=(20, 20, 20), cov_update=40, burn=80, tskip=10, seed=19, rs=d, cov_mode=cov_mode)
    g.run(250)
    o.run(250)
    _compare(g, o, "groups (%s) d=%d " % (cov_mode, d))
    assert_same(g.get("Ut"), o.Ut, "Ut")
    assert_same(g.get("S"), o.S, "S")
    assert o.jstat[..., :3, 0].sum(axis=(0, 1)).min() > 0


@pytest.mark.parametrize("cov_mode,W", [("per_walker", 5), ("per_walker", 70), ("pooled", 6)])
@pytest.mark.parametrize("d,groups", [(6, [[0, 1, 2], [3, 4, 5]]), (5, [[0, 1, 2, 3, 4], [3, 1], [2]]), (40, [list(range(0, 40, 2)), list(range(1, 40, 2)), [7]]),
                                      (100, [list(range(100)), list(range(0, 50)), list(range(30, 100)), [99, 3]])])
def test_parameter_groups_with_the_device_eigensolver(mods, d, groups, cov_mode, W):
    """Parameter groups (PTMCMCSampler.py:129-145, 797-803: one SVD per group's block of the covariance) with eig_mode="ql": every
    covariance epoch factorizes each group's block on the device -- gathered in ascending parameter order, ptmi_eig_ql's kernels
    (one kernel per matrix, or reduce -> chains -> apply from 64 matrices on), the vectors embedded in the full space -- bit for bit
    the oracle's orc_eig_ql on the same blocks.  What real PTA runs combine: groups, per-walker covariances, no host round trip."""
    g, o = _pair(mods, d, 3, W, groups=groups, weights=(20, 20, 20), cov_update=40, burn=80, tskip=10, seed=17, rs=d + 1, cov_mode=cov_mode, eig_mode="ql")
    for n in (45, 120, 85):
        g.run(n)
        o.run(n)
        _compare(g, o, "groups + ql d=%d it=%d " % (d, g.iter))
        assert_same(g.get("Ut"), o.Ut, "Ut it=%d" % g.iter)
        assert_same(g.get("S"), o.S, "S it=%d" % g.iter)
    assert g.eig_epochs >= 6
    assert o.jstat[..., :3, 0].sum(axis=(0, 1)).min() > 0


@pytest.mark.parametrize("d", [641, 1000])
def test_box_prior_fast_path_in_a_narrow_box_far_from_the_origin(mods, d):
    """The 64-lane SCAM kernel's box-prior fast path (mh_steps_kernel BOXFAST: a tracked lower bound of the chain's distance to its
    nearest bound spares most steps the full test, PTMCMCSampler.py:605-606) where its slack is thinnest: bounds [1e6 - 0.5, 1e6 + 0.5],
    so an accepted x + dq rounds at |x| 2^-53 = 1e-10 -- far above the relative slack of a margin of 0.3 -- and the tracked margin
    carries an absolute guard for it (2^-52 x the largest |bound| per accepted step).  Very hot ranks, so that nearly every proposal
    inside the box is accepted and the walls are hit: HIP == oracle bit for bit, refusals by the prior included."""
    orc, _lib, PTEngine = mods
    rs = np.random.RandomState(d)
    lo, hi = 1e6 - 0.5 - 0.05 * rs.rand(d), 1e6 + 0.5 + 0.05 * rs.rand(d)
    g, o = _pair(mods, d, 2, 3, weights=(20, 0, 0), cov_update=60, burn=1000, tskip=25, seed=d, cov_mode="pooled", ladder=[1e14, 1e15],
                 logp=("box", lo, hi), p0=1e6 + rs.uniform(-0.3, 0.3, (3, 2, d)), cov0=np.eye(d) * 0.02)
    for n in (130, 75, 200):
        g.run(n)
        o.run(n)
        _compare(g, o, "narrow far box d=%d it=%d " % (d, g.iter))
    flags, G, E = g.last_variant()
    assert G == 64 and flags & _lib.VAR_UTPAD
    js = o.jstat[..., 0, :].astype(np.int64)
    assert (js[..., 1] > 0.3 * js[..., 0]).all() and (js[..., 1] < js[..., 0]).all()        # mostly accepted, and the walls refused some
    X = g.get("X")
    assert (X >= lo).all() and (X <= hi).all()
