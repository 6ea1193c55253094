"""Runs whose eigenvector table is made ON THE DEVICE by a solver the oracle does not restate (``eig_mode="sytrd"``: the
in-house tridiagonalization + divide and conquer, ``"hipsolver"``: the ROCm library) against the oracle all the same.

The last bits of such a table are the device solver's own, so the table cannot be compared with a LAPACK one.  Everything else can:
the table each launch READS is tapped on the engine's stream right ahead of the launch and handed to the oracle, which then
(a) checks it for what it must be -- an orthonormal U and an S with U diag(S) U^T = the covariance the ORACLE's pooled statistics
made of the same rank-0 rows, to 1e-12 -- at the moment the oracle's own schedule (``eig_lag`` launches after the epoch, or at the
next epoch, PTMCMCSampler.py:545-560 with the engine's late table) puts a new table into force, and (b) steps its chains with it:
proposals (PTMCMCSampler.py:820-985), accept test (:605-622), swaps (:631-697) and the next period's statistics (:769-803) bit for
bit.  A table that came a launch early or late, or was read half-written (round 5's race in the tridiagonalization's barrier),
shows as a chain difference.

Also here: the two bench legs that had no full-size comparison (the replica mode, per-walker covariance + device QL at 64 x 4096;
the dense default mix with one pick per walker at 64 x 4096).

Run on the GPU box: ``python -m pytest tests -m gpu``.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from test_gpu_bench_kernels import _Subset, _check_subset, _dense
from test_gpu_parity import _compare, assert_same, mods  # noqa: F401  (mods is a fixture)

pytestmark = pytest.mark.gpu

TOL = 1e-12
RESIDUALS = []         # (what, ndim, |U diag(S) U^T - cov|_max / |cov|_max, |U U^T - I|_max) of every table checked (printed with pytest -s)


@pytest.fixture(scope="module", autouse=True)
def _report_residuals():
    yield
    if RESIDUALS:
        worst = {}
        for what, d, rc, ro in RESIDUALS:
            w = worst.setdefault(d, [0, 0.0, 0.0])
            w[0], w[1], w[2] = w[0] + 1, max(w[1], rc), max(w[2], ro)
        print("\ndevice tables checked against the oracle's covariances (ndim: tables, worst |U S U^T - cov| / |cov|, worst |U U^T - I|): "
              + "; ".join("%d: %d, %.2e, %.2e" % ((d,) + tuple(v)) for d, v in sorted(worst.items())))


class TableTap(object):
    """Every fused launch of ``g`` with the table it reads: (iter0, nsteps, Ut, S), the copies queued on the engine's stream right
    ahead of the launch (whatever ``run`` put into force before it is in them, whatever it does behind the launch is not)."""

    def __init__(self, g):
        self.g, self.launches = g, []
        orig = g.mh_steps

        def tapped(iter0, nsteps):
            self.launches.append((int(iter0), int(nsteps), g.t["Ut"].clone(), g.t["S"].clone()))
            orig(iter0, nsteps)

        g.mh_steps = tapped

    def take(self):
        out, self.launches = [(i, n, u.cpu().numpy(), s.cpu().numpy()) for i, n, u, s in self.launches], []
        return out


def check_table(Ut, S, cov, what):
    """Ut [d][d] (eigenvectors as rows), S [d] against the covariance they were made from."""
    d = len(S)
    scale = np.abs(cov).max()
    rec = (Ut.T * S) @ Ut
    r_cov, r_orth = np.abs(rec - cov).max() / scale, np.abs(Ut @ Ut.T - np.eye(d)).max()
    RESIDUALS.append((what, d, r_cov, r_orth))
    assert r_cov <= TOL, "%s: U diag(S) U^T misses the covariance by %.3g of its largest entry" % (what, r_cov)
    assert r_orth <= TOL, "%s: U is not orthonormal (%.3g)" % (what, r_orth)
    assert (S >= 0).all() and (np.diff(S) <= 0).all(), "%s: S is not sorted by decreasing size" % what


def lockstep(g, o, tap, n, what):
    """``g.run(n)``, then ``o.run(n)`` with every table ``o``'s schedule puts into force taken from the device: the next distinct
    table the launches of ``g`` read (or hold behind the last launch).  ``o`` is an OracleEngine with the engine's ``eig_lag``; its
    factorization is replaced, its statistics and schedule are its own."""
    g.run(n)
    g.sync()
    seq = [(u, s) for _, _, u, s in tap.take()] + [(g.get("Ut"), g.get("S"))]
    queue = []
    last = (o.Ut.copy(), o.S.copy())
    for u, s in seq:
        if not (np.array_equal(u, last[0]) and np.array_equal(s, last[1])):
            queue.append((u, s))
            last = (u, s)
    used = [0]

    def from_device(w):
        assert w == 0 and queue, "%s: the oracle's schedule puts a table into force at iteration <= %d that the device never used" % (what, o.iter + n)
        u, s = queue.pop(0)
        for gi, grp in enumerate(o.groups):         # a group's table: its vectors on its own parameters against its block of the covariance
            m = len(grp)
            check_table(u[0, gi][:m][:, grp], s[0, gi][:m], o.cov[0][np.ix_(grp, grp)], "%s, table %d, group %d" % (what, g.eig_epochs - len(queue), gi))
            if len(o.groups) > 1:
                outside = np.setdiff1d(np.arange(o.d), grp)
                assert not u[0, gi][:, outside].any() and not u[0, gi][m:].any() and not s[0, gi][m:].any()      # embedded: zero elsewhere
        o.Ut[...], o.S[...] = u, s
        used[0] += 1

    o._svd = from_device
    o.run(n)
    assert not queue, "%s: the device used %d table(s) the oracle's schedule does not know" % (what, len(queue))
    return used[0]


@pytest.mark.parametrize("eig_mode,d,lag,tskip,weights", [
    ("sytrd", 300, 0, 10, (20, 0, 0)),             # on the engine's stream, at once
    ("sytrd", 300, 2, 10, (20, 0, 0)),             # on the side stream, two launches late
    ("sytrd", 300, 5, 10, (20, 20, 20)),           # lag > the period's three launches: finished behind the next epoch's statistics; AM + DE read the table too
    ("sytrd", 130, 1, 8, (20, 20, 0)),             # 16 lanes per chain
    ("sytrd", 1000, 3, 10, (20, 0, 0)),            # the 64-lane kernel over the padded copy of a device-made table
    ("hipsolver", 300, 2, 10, (20, 0, 0)),
    ("hipsolver", 200, 0, 10, (20, 20, 0)),
])
def test_device_factorized_run_equals_the_oracle_on_the_device_tables(mods, eig_mode, d, lag, tskip, weights):
    """Small batches, the WHOLE run on the oracle (its own pooled statistics from its own rows)."""
    orc, _lib, PTEngine = mods
    nt, W, cu = 4, 6, 30
    kw = dict(weights=weights, cov_update=cu, burn=2 * cu, tskip=tskip, seed=19, cov_mode="pooled", eig_lag=lag, cov0=np.eye(d) * 0.01)
    p0 = np.random.RandomState(d).randn(W, nt, d) * 0.3
    cov0 = kw.pop("cov0")
    g = PTEngine(d, nt, W, cov0, eig_mode=eig_mode, **kw)
    o = orc.OracleEngine(d, nt, W, cov0, **kw)         # the oracle knows none of the device solvers: its factorization becomes the hook
    g.init_state(p0)
    o.init_state(p0)
    assert g.eig_lag == lag and o.eig_lag == lag
    tap = TableTap(g)
    used = 0
    for n in (cu + tskip, 2 * cu, 7, 3 * cu + 3):
        used += lockstep(g, o, tap, n, "%s d=%d lag=%d it=%d" % (eig_mode, d, lag, g.iter + n))
        _compare(g, o, "%s d=%d lag=%d it=%d " % (eig_mode, d, lag, g.iter))
        assert_same(g.get("cov"), o.cov, "cov it=%d" % g.iter)
    assert used >= 5 and g.eig_epochs >= 5
    if sum(weights[1:]):
        assert o.jstat[..., 1, 1].sum() > 0


def _lag_schedule(n_launch_per_period, lag):
    """Index (0-based, counted from the epoch) of the first launch that reads the epoch's table: ``lag`` launches late, at the
    latest the first launch of the next period (a new epoch finishes a pending factorization first)."""
    return min(lag, n_launch_per_period)


@pytest.mark.parametrize("tskip,lag,eig_mode", [(100, 10, "sytrd"), (20, 3, "sytrd")])
def test_config4_share_as_benchmarked_sytrd_late_table(mods, tskip, lag, eig_mode):
    """BASELINE configs[3], one GPU's share AS bench.py times it (``also.config4_share_1000d_64x512``: cov_mode pooled_sytrd, eig_lag
    10, am_mode rle): 64 ranks x 512 walkers x 1000-d, SCAM cycle, the 64-lane kernel over the padded copy of a table that
    ptmi_eig_sytrd_from made on the side stream BESIDE the launches.  covUpdate = 100, 400 iterations = three covariance epochs.
    ``tskip=100, lag=10``: a period is one launch, every table is finished behind the NEXT epoch's statistics (the bench's order at
    its own period of ten launches); ``tskip=20, lag=3``: the table takes effect three launches into its period.  The device's
    table of every epoch decomposes the ORACLE's pooled covariance of the same rows to 1e-12, comes into force at the launch the lag
    rule names, and three walkers stepped with it equal the device's bit for bit throughout."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 1000, 64, 512, 100
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=10000, tskip=tskip, seed=5)
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", eig_mode=eig_mode, eig_lag=lag, am_mode="rle", **kw)
    assert g.am_rle and g.eig_lag == lag and g.late_finish
    g.init_state(np.zeros(d))
    tap = TableTap(g)
    sub = _Subset(orc, (0, 300, 511), d, nt, W, cov0, am_mode="rle", **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
    per = cu // tskip                                  # launches per covariance period
    covs = []                                          # the oracle's covariance of every epoch
    in_force, cur = 0, (sub.subs[0].Ut.copy(), sub.subs[0].S.copy())      # number of device tables that have come into force; the one in force
    changes = []
    for k in range(4):
        if k > 0:
            g.sync()
            sub.epoch(g.get("AM"), k * cu, g.get("AMflag"), apply=False)
            covs.append(sub.subs[0].cov[0].copy())
        g.run(cu)                                      # starts with the device's own epoch
        g.sync()
        if k > 0:
            assert_same(g.get("cov")[0], covs[-1], "pooled cov of epoch %d" % k)
        for j, (it0, ns, u, s) in enumerate(tap.take()):
            assert ns == tskip and it0 == k * cu + j * tskip + 1
            if not (np.array_equal(u, cur[0]) and np.array_equal(s, cur[1])):
                check_table(u[0, 0], s[0, 0], covs[in_force], "table of epoch %d" % (in_force + 1))
                in_force += 1
                cur = (u, s)
                changes.append((k, j))
                assert (np.abs(u[0, 0]) > 1e-9).mean() > 0.9        # an adapted, dense table
            for o in sub.subs:
                o.Ut[...], o.S[...] = cur
                o.run(ns)
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_UTPAD and not flags & _lib.VAR_FULL and (G, E) == (64, 16)
        _check_subset(g, sub, "config 4 share, device tables, period %d" % k)
    # where the lag rule puts every epoch's table: epoch e (end of period e - 1) -> launch min(lag, per) of period e
    first = _lag_schedule(per, lag)
    want = [(e + first // per, first % per) for e in range(1, 4)]
    want = [c for c in want if c[0] < 4]
    assert changes == want, "tables came into force at (period, launch) %r, the lag rule says %r" % (changes, want)
    assert g.eig_epochs >= len(want)
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * (X ** 2).sum(-1), rtol=1e-12, atol=1e-12)
    assert g.get("nswap").sum() > 0 and g.swap_proposed == 400 // tskip


def test_full_size_replica_mode_per_walker_device_ql(mods):
    """The bench leg ``config2_replica_per_walker_cov_device_ql`` at its own size: 64 ranks x 4096 walkers x 100-d, SCAM cycle,
    EVERY walker a replica of a reference run with its own covariance, eigenvectors (ptmi_eig_ql: 4096 device factorizations per
    epoch) and rank-0 ring (PTMCMCSampler.py:769-803, 820-876).  Walkers are independent here, so three of them run on the oracle as
    one-walker engines (``walker0`` = their RNG streams, eig_mode "ql" = the restated device solver): chains, covariance, table
    bit for bit through three covariance epochs and four swap epochs."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 100, 64, 4096, 100
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=10000, tskip=100, seed=1234, cov_mode="per_walker", eig_mode="ql")
    g = PTEngine(d, nt, W, cov0, **kw)
    g.init_state(np.zeros(d))
    walkers = (0, 2049, 4095)
    subs = [orc.OracleEngine(d, nt, 1, cov0, walker0=w0, **kw) for w0 in walkers]
    for o in subs:
        o.init_state(np.zeros(d))
    for k in range(4):
        g.run(cu)
        g.sync()
        X, lnL, so, nacc, nsw = g.get("X"), g.get("lnL"), g.get("slot_of"), g.get("nacc"), g.get("nswap")
        cov, Ut, S = g.get("cov"), g.get("Ut"), g.get("S")
        for w0, o in zip(walkers, subs):
            o.run(cu)
            what = "replica walker %d, period %d: " % (w0, k)
            assert_same(X[w0], o.X[0], what + "X")
            assert_same(lnL[w0], o.lnL[0], what + "lnL")
            assert_same(so[w0], o.slot_of[0], what + "slot_of")
            assert_same(nacc[w0], o.nacc[0], what + "nacc")
            assert_same(nsw[w0], o.nswap[0], what + "nswap")
            assert_same(cov[w0], o.cov[0], what + "cov")
            assert_same(Ut[w0], o.Ut[0], what + "Ut")
            assert_same(S[w0], o.S[0], what + "S")
    assert g.eig_epochs == 3 and not np.array_equal(Ut[0], Ut[4095])
    assert (np.abs(Ut[2049, 0]) > 1e-9).mean() > 0.5                      # adapted tables (the start's are unit vectors)
    assert g.get("nswap").sum() > 0 and g.swap_proposed == 4


def test_full_size_dense_default_mix_walker_pick(mods):
    """The bench leg ``config3_dense_default_mix_walker_pick`` at its own size: 100-d dense Gaussian, 64 x 4096 chains, SCAM / AM /
    DE 20 / 20 / 20 with ONE cycle pick per walker and iteration (mh_pc_kernel<25, dense, flat, persistent>: the likelihood's table
    in LDS, stepper and AM-producer waves paired; PTMCMCSampler.py:605-612, 820-985), pooled covariance with covUpdate = 100 and
    burn = 200, 400 iterations: three covariance epochs, a DE epoch and DE activation, four swap epochs.  Three walkers bit for bit
    throughout; pooled cov / Ut / S / DE history equal the oracle's on the same rank-0 rows."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 20, 20), cov_update=100, burn=200, tskip=100, seed=321, logl=_dense(d), pick_mode="walker")
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", **kw)
    g.init_state(np.zeros(d))
    sub = _Subset(orc, (0, 1500, 4095), d, nt, W, cov0, **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
    cu = kw["cov_update"]
    for k in range(4):
        if k > 0:
            g.sync()
            sub.epoch(g.get("AM"), k * cu, g.get("AMflag") if g.am_rle else None)
        g.run(cu)
        for o in sub.subs:
            o.run(cu)
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_PC and flags & _lib.VAR_PERSISTENT and flags & _lib.VAR_UNIFORM and flags & _lib.VAR_FULL
        assert not flags & _lib.VAR_LDS_UT and (G, E) == (4, 25)
        if k > 0:
            assert_same(g.get("cov")[0], sub.subs[0].cov[0], "pooled cov after epoch %d" % k)
            assert_same(g.get("Ut")[0], sub.subs[0].Ut[0], "Ut after epoch %d" % k)
            assert_same(g.get("S")[0], sub.subs[0].S[0], "S after epoch %d" % k)
        if k * cu >= kw["burn"]:
            assert_same(np.roll(g.get("DE")[0], -g.de_head, axis=0), sub.DE, "DE history")
        _check_subset(g, sub, "dense mix, walker pick, segment %d" % k)
    assert g.de_on
    js = g.get("jstat").astype(np.int64)
    assert (js[..., :3, 0].sum(-1) == 400).all() and js[..., 2, 0].sum() > 0 and js[..., 1, 1].sum() > 0
    assert (js[..., :3, 0] == js[:, :1, :3, 0]).all()                    # one pick per walker: the same counts down its ladder
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * np.einsum("wti,ij,wtj->wt", X, kw["logl"][2], X), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("d,groups", [(300, [np.arange(0, 100), np.arange(100, 250), np.arange(250, 300)]),
                                      (40, [np.array([0, 2, 4, 6, 39]), np.arange(7, 39), np.array([1, 3, 5])])])
def test_parameter_groups_with_the_library_eigensolver_on_the_device(mods, d, groups):
    """Parameter groups (PTMCMCSampler.py:129-145, 797-803: one factorization per group's block of the covariance) with
    eig_mode="hipsolver" -- any ndim, where the QL kernels stop at 128: every block gathered on the device, factorized by the ROCm
    library, its vectors embedded in the full space.  The library's last bits are its own: every group's table is checked against the
    ORACLE's covariance block and the oracle's chains are stepped with it (AM increments ahead of the launch group by group, DE's
    masks): bit for bit."""
    orc, _lib, PTEngine = mods
    nt, W, cu = 3, 5, 30
    kw = dict(weights=(20, 20, 20), cov_update=cu, burn=2 * cu, tskip=10, seed=5, cov_mode="pooled", groups=groups)
    cov0 = np.eye(d) * 0.01
    p0 = np.random.RandomState(d).randn(W, nt, d) * 0.2
    g = PTEngine(d, nt, W, cov0, eig_mode="hipsolver", **kw)
    o = orc.OracleEngine(d, nt, W, cov0, **kw)
    g.init_state(p0)
    o.init_state(p0)
    tap = TableTap(g)
    used = 0
    for n in (cu + 10, 2 * cu, 7, 2 * cu + 3):
        used += lockstep(g, o, tap, n, "groups + hipsolver d=%d it=%d" % (d, g.iter + n))
        _compare(g, o, "groups + hipsolver d=%d it=%d " % (d, g.iter))
        assert_same(g.get("cov"), o.cov, "cov it=%d" % g.iter)
    assert used >= 4 and o.jstat[..., :3, 0].sum(axis=(0, 1)).min() > 0 and o.jstat[..., 1, 1].sum() > 0
