"""AM row flags (include/ptmi.h ``ptmi_buffers.AMflag``, ``PTEngine(am_mode="rle")``): a rejected proposal leaves the rank-0 chain
where it was, so the step kernels store its row (updateChains' buffer, PTMCMCSampler.py:327-328) only when the step was accepted
or the row is a KEY row, and the pooled statistics take every stored row once, weighted by its run length
(``orc_pool_update_rle``).  Everything a reader sees -- every row of the ring (``ptmi_am_expand``), the pooled covariance, the
chains -- equals the oracle bit for bit; and the weighted sums are the plain ones up to rounding.

Run on the GPU box: ``python -m pytest tests -m gpu``.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from test_gpu_parity import _compare, _pair, assert_same, mods  # noqa: F401  (mods is a fixture)

pytestmark = pytest.mark.gpu


def _dense(d, seed=0):
    A = np.random.default_rng(seed).standard_normal((d, d))
    return ("dense", np.zeros(d), np.linalg.inv(A @ A.T / d + np.eye(d)))


CASES = [
    # d, nt, W, cov_update, tskip, weights, extra
    (100, 64, 3, 40, 20, (20, 0, 0), {}),                       # the persistent kernel of the exact shape, swaps, epochs
    (100, 5, 9, 25, 7, (20, 0, 0), {}),                         # a ring shorter than a chunk, a swap period that does not divide it
    (100, 1, 6, 30, 0, (20, 0, 0), {}),                         # one temperature: no swap ever writes a KEY row
    (100, 64, 2, 50, 25, (20, 0, 0), {"logl": "dense"}),        # mh_dense_scam_kernel
    (100, 8, 4, 33, 11, (20, 0, 0), {"box": True}),             # box prior: many rejected proposals, long runs
    (100, 64, 2, 40, 20, (20, 20, 20), {"burn": 80}),           # the staged default-mix kernel, DE history filled from expanded rows
    (37, 6, 5, 20, 10, (20, 20, 0), {}),                        # a 4-lane shape with padding slots, AM in the cycle
    (7, 3, 4, 16, 4, (20, 0, 0), {}),
    (130, 4, 3, 24, 8, (20, 0, 0), {}),                         # 16 lanes per chain, two macro tiles in the statistics
    (500, 2, 2, 16, 8, (20, 0, 20), {"burn": 32}),              # 64 lanes per chain, DE
    (20, 4, 5, 30, 10, (10, 0, 10), {"nuts": True}),            # the gradient-jump kernel
    (100, 8, 3, 40, 20, (20, 20, 20), {"burn": 60}),            # burn no multiple of covUpdate: a DE epoch reads rows of the period before
    (100, 8, 3, 40, 10, (20, 0, 20), {"burn": 25}),             # burn < covUpdate: the DE history takes the ring's last 25 rows
    (37, 4, 3, 30, 10, (20, 20, 20), {"burn": 70}),
]


@pytest.mark.parametrize("d,nt,W,cu,tskip,weights,extra", CASES)
def test_rle_engine_matches_the_oracle(mods, d, nt, W, cu, tskip, weights, extra):
    orc, _lib, PTEngine = mods
    kw = dict(weights=weights, cov_update=cu, burn=extra.get("burn", 1000), tskip=tskip, seed=31, cov_mode="pooled", cov0=np.eye(d) * 0.01)
    if extra.get("logl") == "dense":
        kw["logl"] = _dense(d)
    if extra.get("box"):
        rs = np.random.RandomState(4)
        kw["logp"] = ("box", -0.25 - rs.rand(d) * 0.1, 0.2 + rs.rand(d) * 0.1)
        kw["p0"] = rs.uniform(-0.05, 0.05, (W, nt, d))
    if extra.get("nuts"):
        kw.update(logl=("curved",), logp=("box", -10 * np.ones(d), 10 * np.ones(d)), cov0=np.eye(d), grad_weights=(10, 5),
                  p0=np.tile(np.array([-0.1, -0.5] * (d // 2)), (W, nt, 1)))
    g, o = _pair(mods, d, nt, W, am_mode="rle", **kw)
    assert g.am_rle and o.am_rle
    total = 0
    for n in (cu + 3, 1, 2 * cu - 5, 17, cu):                   # launches of odd lengths, epochs inside
        g.run(n)
        o.run(n)
        total += n
        _compare(g, o, "rle d=%d it=%d " % (d, total))          # get("AM") copies the repeats of the current period forward
        lo, hi = g.am_period()
        rows = np.arange(lo, hi + 1) % cu
        assert_same(g.get("AMflag")[:, rows] & 3, o.AMflag[:, rows] & 3, "flags d=%d it=%d" % (d, total))
        assert_same(g.get("cov"), o.cov, "cov d=%d it=%d" % (d, total))
        assert_same(g.get("Ut"), o.Ut, "Ut")
    fl = g.get("AMflag")
    assert 0 < ((fl & 3) == 0).mean()                           # some rows were not stored
    if kw["burn"] < total:
        assert_same(np.roll(g.get("DE")[0], -g.de_head, axis=0), o.DE[0], "DE history")


def test_rows_mode_still_matches_its_oracle_and_rle_within_rounding(mods):
    """am_mode="rows" keeps the plain sums (orc_pool_update); the weighted sums of am_mode="rle" are the same statistics up to
    rounding: covariances agree to 1e-12 relative after the first epoch (the runs then part at the level of the last bits)."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 100, 8, 6, 60
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=20, seed=9, cov_mode="pooled", cov0=np.eye(d) * 0.01)
    r, ro = _pair(mods, d, nt, W, am_mode="rows", **kw)
    g, go = _pair(mods, d, nt, W, am_mode="rle", **kw)
    assert not r.am_rle and r.t["AMflag"] is None and not ro.am_rle
    r.run(cu + 1)
    ro.run(cu + 1)
    g.run(cu + 1)
    _compare(r, ro, "rows ")
    assert_same(r.get("cov"), ro.cov, "rows cov")
    a, b = r.get("cov")[0], g.get("cov")[0]
    assert not np.array_equal(a, b) and np.abs(a - b).max() < 1e-12 * np.abs(a).max()
    r.run(3 * cu)
    ro.run(3 * cu)
    _compare(r, ro, "rows later ")


def test_expand_of_a_range_leaves_the_other_rows_alone(mods):
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 100, 4, 6, 50
    rs = np.random.RandomState(4)
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=10, seed=5, cov_mode="pooled", cov0=np.eye(d) * 0.01,
              logp=("box", -0.25 - rs.rand(d) * 0.1, 0.2 + rs.rand(d) * 0.1), p0=rs.uniform(-0.05, 0.05, (W, nt, d)))
    g, o = _pair(mods, d, nt, W, am_mode="rle", **kw)
    g.run(130)
    o.run(130)
    g.sync()
    before = g.am_params(g.t["AM"].cpu().numpy())               # stored rows valid, the repeats stale
    g.am_expand(2, 3, 111, 125)                                 # walkers 2..4, iterations 111..125
    after = g.am_params(g.t["AM"].cpu().numpy())
    rows = np.arange(111, 126) % cu
    assert_same(after[2:5][:, rows], o.AM[2:5][:, rows], "expanded range")
    mask = np.ones(after.shape[:2], bool)
    mask[2:5, rows] = False
    assert_same(after[mask], before[mask], "rows outside the range")
    assert not np.array_equal(before[2:5][:, rows], o.AM[2:5][:, rows])      # they did need filling
    with pytest.raises(_lib.PtmiError):
        g.am_expand(0, 1, 90, 125)                              # reaches into the period before: refused


def test_rle_is_refused_for_per_walker_covariances(mods):
    orc, _lib, PTEngine = mods
    with pytest.raises(ValueError):
        PTEngine(10, 2, 2, np.eye(10), weights=(20, 0, 0), cov_mode="per_walker", am_mode="rle")
    g = PTEngine(10, 2, 2, np.eye(10), weights=(20, 20, 20), cov_mode="per_walker")      # auto: rows
    assert not g.am_rle and g.t["AMflag"] is None


def test_checkpoint_of_an_rle_run_continues_bit_identically(mods):
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 8, 5
    kw = dict(weights=(20, 0, 0), cov_update=40, burn=1000, tskip=10, seed=77, cov_mode="pooled")
    a = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    a.init_state(np.zeros(d))
    a.run(95)
    st = a.checkpoint()
    a.run(130)
    b = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    b.init_state(np.zeros(d))
    b.restore(st)
    b.run(130)
    for name in ("X", "lnL", "cov", "Ut", "S"):
        assert_same(a.get(name), b.get(name), name)
    lo, hi = a.am_period()
    rows = np.arange(lo, hi + 1) % 40
    assert_same(a.get("AM")[:, rows], b.get("AM")[:, rows], "AM")


@pytest.mark.parametrize("am_mode,weights", [("rle", (20, 0, 0)), ("rows", (20, 0, 0)), ("rle", (20, 20, 20))])
def test_eig_lag_applies_the_table_one_launch_late(mods, am_mode, weights):
    """eig_lag = 1 (pooled covariance, host factorization): the launch that follows a covariance epoch still runs with the table in
    force, the host factorizes meanwhile, the launch after that uses the result.  HIP == oracle (OracleEngine(eig_lag=1)) bit for bit;
    against eig_lag = 0 the chains are the same up to the end of the epoch's first launch and differ afterwards."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu, tskip = 100, 8, 5, 40, 10
    kw = dict(weights=weights, cov_update=cu, burn=80, tskip=tskip, seed=13, cov_mode="pooled", cov0=np.eye(d) * 0.01, am_mode=am_mode)
    g, o = _pair(mods, d, nt, W, eig_lag=1, **kw)
    z, _ = _pair(mods, d, nt, W, eig_lag=0, **kw)
    assert g.eig_lag == 1 and o.eig_lag == 1
    for n in (cu, tskip, 3, cu - 3, 2 * cu + 7, 33):
        before = g.get("Ut").copy()
        g.run(n)
        o.run(n)
        z.run(n)
        _compare(g, o, "lag it=%d " % g.iter)
        assert_same(g.get("Ut"), o.Ut, "Ut it=%d" % g.iter)
        assert_same(g.get("cov"), o.cov, "cov it=%d" % g.iter)
        if g.iter == cu:
            assert np.array_equal(g.get("X"), z.get("X"))                # nothing adapted yet
        if g.iter == cu + tskip:
            assert not np.array_equal(before, g.get("Ut"))               # the table changed during this run ...
            assert np.array_equal(g.get("Ut"), z.get("Ut"))              # ... into the same table (same rows, same statistics) ...
            assert not np.array_equal(g.get("X"), z.get("X"))            # ... but this launch still ran with the old one
    assert not np.array_equal(g.get("X"), z.get("X"))
    assert g.eig_epochs == z.eig_epochs


@pytest.mark.parametrize("lag", [2, 3, 4, 7])
def test_eig_lag_of_several_launches(mods, lag):
    """eig_lag = L launches (four launches per covariance period here): the table of an epoch takes effect L launches later, at the
    latest at the next epoch (L >= 4: its statistics are then taken before it is in force).  HIP == OracleEngine(eig_lag=L)."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu, tskip = 100, 8, 5, 40, 10
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=tskip, seed=21, cov_mode="pooled", cov0=np.eye(d) * 0.01)
    g, o = _pair(mods, d, nt, W, eig_lag=lag, **kw)
    for n in (cu + tskip, 7, 3 * cu - 7, 2 * cu):
        g.run(n)
        o.run(n)
        _compare(g, o, "lag %d it=%d " % (lag, g.iter))
        assert_same(g.get("Ut"), o.Ut, "Ut it=%d" % g.iter)
    assert g.eig_epochs >= 5


@pytest.mark.parametrize("lag", [1, 2, 3])
def test_checkpoint_between_an_epoch_and_its_late_table(mods, lag):
    """A checkpoint taken while a factorization is pending (eig_lag: between a covariance epoch and the launch its table takes effect
    at) carries the pending state; the restored run applies the table at the same launch: bit-identical continuation."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu, tskip = 100, 8, 5, 40, 10
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=tskip, seed=3, cov_mode="pooled", eig_lag=lag)
    a = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    a.init_state(np.zeros(d))
    a.run(cu + tskip)                                             # the epoch at iteration cu + 1, one launch behind it
    assert a._eig_pending == (lag > 1)
    a.run(cu)                                                     # ... and again one launch behind the next epoch
    st = a.checkpoint()
    assert st["eig_pending"] == int(lag > 1)
    ut_then = a.get("Ut").copy()
    a.run(2 * cu + 7)
    b = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    b.init_state(np.zeros(d))
    b.restore(st)
    assert_same(b.get("Ut"), ut_then, "table in force at the checkpoint")
    b.run(2 * cu + 7)
    for name in ("X", "lnL", "cov", "Ut", "S", "nacc"):
        assert_same(a.get(name), b.get(name), name)
    assert a.eig_epochs == b.eig_epochs


def test_a_failed_side_stream_factorization_is_reported(mods, monkeypatch):
    """eig_mode="hipsolver" with eig_lag: the library's eigensolver runs on a host thread of its own; an exception there must stop the
    run when its table is due, not leave the sampler with an unwritten table."""
    import torch
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 100, 4, 3, 20
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=10, seed=1, cov_mode="pooled",
                 eig_mode="hipsolver", eig_lag=1)
    g.init_state(np.zeros(d))
    g.run(cu)
    before = g.get("Ut").copy()

    def boom(*a, **k):
        raise RuntimeError("injected")

    monkeypatch.setattr(torch.linalg, "eigh", boom)
    with pytest.raises(_lib.PtmiError, match="side stream"):
        g.run(cu)
    g.sync()
    assert_same(g.get("Ut"), before, "the table in force is untouched")


ASYNC_CASES = [
    # d, nt, W, cu, tskip, weights, burn, lag, am_mode
    (100, 64, 3, 40, 20, (20, 0, 0), 1000, 1, "rle"),            # the persistent kernel, one launch of lag
    (100, 8, 5, 40, 10, (20, 0, 0), 1000, 3, "rows"),            # every row stored; the table three launches late
    (100, 8, 4, 40, 10, (20, 20, 20), 80, 2, "rle"),             # DE history: the DE epoch reads the finished period's ring before the switch
    (100, 8, 4, 30, 10, (20, 20, 20), 60, 5, "rows"),            # lag longer than a period (three launches): the next epoch finishes it first
    (130, 4, 3, 24, 8, (20, 0, 0), 1000, 1, "rle"),              # 16 lanes per chain, the padded table copy
    (37, 6, 5, 20, 10, (20, 20, 0), 1000, 2, "rle"),
]


@pytest.mark.parametrize("d,nt,W,cu,tskip,weights,burn,lag,am_mode", ASYNC_CASES)
def test_statistics_on_the_side_stream_change_nothing(mods, d, nt, W, cu, tskip, weights, burn, lag, am_mode):
    """stats_async: two AM rings, the statistics of a finished covariance period and the factorization on a side stream beside the
    launches of the next one (PTMCMCSampler.py:545-560 with the table eig_lag launches late).  A scheduling change only: chains,
    covariance, table, DE history and every row a reader of the ring sees equal OracleEngine(eig_lag=L) bit for bit."""
    orc, _lib, PTEngine = mods
    kw = dict(weights=weights, cov_update=cu, burn=burn, tskip=tskip, seed=41, cov_mode="pooled", cov0=np.eye(d) * 0.01, am_mode=am_mode, eig_lag=lag)
    g, o = _pair(mods, d, nt, W, stats_async=True, **{k: v for k, v in kw.items()})
    assert g.stats_async and g.eig_lag == lag
    for n in (cu, tskip, 3, cu - 3, 2 * cu + 7, 33, cu):
        g.run(n)
        o.run(n)
        _compare(g, o, "async it=%d " % g.iter)                   # incl. the ring as a reader sees it (the two rings merged)
        assert_same(g.get("Ut"), o.Ut, "Ut it=%d" % g.iter)
        assert_same(g.get("cov"), o.cov, "cov it=%d" % g.iter)
        if burn < g.iter:
            assert_same(np.roll(g.get("DE")[0], -g.de_head, axis=0), o.DE[0], "DE history it=%d" % g.iter)
    assert g.eig_epochs >= 4


def test_statistics_on_the_side_stream_with_the_device_factorization(mods):
    """stats_async with eig_mode="sytrd" (config 4's combination: statistics, tridiagonalization and the library's solver all on the
    side stream): the same run as with everything on the engine's stream, bit for bit (no oracle for the library's last bits)."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 300, 4, 6, 30
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=10, seed=8, cov_mode="pooled", eig_mode="sytrd", eig_lag=2)
    runs = []
    for asy in (False, True):
        g = PTEngine(d, nt, W, np.eye(d) * 0.01, stats_async=asy, **kw)
        g.init_state(np.zeros(d))
        for n in (cu + 10, 2 * cu, 7, 3 * cu):
            g.run(n)
        g.sync()
        runs.append({k: g.get(k) for k in ("X", "lnL", "cov", "Ut", "S", "nacc")})
        assert g.eig_epochs >= 5
    for k in runs[0]:
        assert_same(runs[0][k], runs[1][k], k)


def test_checkpoint_of_a_run_with_statistics_on_the_side_stream(mods):
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 100, 8, 5, 40
    kw = dict(weights=(20, 20, 20), cov_update=cu, burn=80, tskip=10, seed=12, cov_mode="pooled", eig_lag=2, stats_async=True)
    a = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    a.init_state(np.zeros(d))
    a.run(2 * cu + 10)                                            # one launch behind an epoch: its table is pending
    st = a.checkpoint()
    assert st["eig_pending"] == 1
    a.run(3 * cu + 5)
    b = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    b.init_state(np.zeros(d))
    b.restore(st)
    b.run(3 * cu + 5)
    for name in ("X", "lnL", "cov", "Ut", "S", "nacc", "DE"):
        assert_same(a.get(name), b.get(name), name)


@pytest.mark.parametrize("eig_mode,lag", [("sytrd", 3), ("sytrd", 5), ("hipsolver", 3)])
def test_pending_device_factorization_is_finished_behind_the_next_statistics(mods, eig_mode, lag):
    """eig_lag >= the launches of a covariance period (three here) with a device factorization: the table of epoch E is still pending
    when epoch E + 1 arrives; the engine queues that epoch's statistics FIRST (they do not read the table) and waits for the old
    table behind them (PTEngine.late_finish), so the factorization has the statistics' time on top of the period's launches.  The
    same table in force for the same launches: bit for bit the run that finishes it before the statistics."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 300, 4, 6, 30
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=10, seed=19, cov_mode="pooled", eig_mode=eig_mode, eig_lag=lag)
    runs = []
    for late in (True, False):
        g = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
        assert g.late_finish
        g.late_finish = late
        g.init_state(np.zeros(d))
        for n in (cu + 10, 2 * cu, 7, 3 * cu + 3):
            g.run(n)
        g.sync()
        runs.append({k: g.get(k) for k in ("X", "lnL", "cov", "Ut", "S", "nacc")})
        assert g.eig_epochs >= 5
    for k in runs[0]:
        assert_same(runs[0][k], runs[1][k], k)


@pytest.mark.parametrize("d,lag", [(300, 2), (1000, 3)])
def test_a_device_factorization_beside_step_launches_is_repeatable(mods, d, lag):
    """The tridiagonalization's grid barrier beside the wide step kernels (eig_lag > 0: the factorization runs on the side stream
    while the chains step): each block's exchanged vectors have to be acknowledged before the barrier's counter moves.  Without the
    wait the same run differed from repeat to repeat (another block read a vector's old contents whenever the step launches kept
    the memory path busy): three repeats, bit for bit."""
    orc, _lib, PTEngine = mods
    nt, W, cu = 4, 6, 30
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=10, seed=8, cov_mode="pooled", eig_mode="sytrd", eig_lag=lag)
    runs = []
    for rep in range(3):
        g = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
        g.init_state(np.zeros(d))
        snaps = []
        for n in (cu + 10, 2 * cu, 7, 3 * cu):
            g.run(n)
            g.sync()
            snaps.append({k: g.get(k).copy() for k in ("X", "Ut", "S", "cov")})
        runs.append(snaps)
        del g
    for rep in (1, 2):
        for i, (sa, sb) in enumerate(zip(runs[0], runs[rep])):
            for k in sa:
                assert_same(sa[k], sb[k], "repeat %d, snapshot %d, %s" % (rep, i, k))


@pytest.mark.parametrize("cov_mode,lag", [("per_walker", 1), ("per_walker", 2), ("per_walker", 3), ("per_walker", 6), ("pooled", 2)])
def test_device_ql_on_the_side_stream_with_eig_lag(mods, cov_mode, lag):
    """eig_mode="ql" with eig_lag (round 5): every walker's covariance (the replica mode: a walker IS a reference run,
    PTMCMCSampler.py:769-803), or the pooled one, is factorized by ptmi_eig_ql_from on a side stream beside the launches that follow
    the epoch; the tables take effect L launches later -- three launches per covariance period here, so lag 3 and 6 finish the
    pending factorization behind the next epoch's statistics.  HIP == OracleEngine(eig_lag=L, eig_mode="ql") bit for bit,
    chains, covariances and tables."""
    orc, _lib, PTEngine = mods
    d, nt, W, cu, tskip = 20, 4, 6, 30, 10
    kw = dict(weights=(20, 20, 20), cov_update=cu, burn=60, tskip=tskip, seed=23, cov_mode=cov_mode, eig_mode="ql", cov0=np.eye(d) * 0.01)
    g, o = _pair(mods, d, nt, W, eig_lag=lag, **kw)
    assert g.eig_lag == lag and o.eig_lag == lag and g.late_finish
    for n in (cu + tskip, 7, 3 * cu - 7, 2 * cu + 3):
        g.run(n)
        o.run(n)
        _compare(g, o, "%s lag %d it=%d " % (cov_mode, lag, g.iter))
        assert_same(g.get("cov"), o.cov, "cov it=%d" % g.iter)
        assert_same(g.get("Ut"), o.Ut, "Ut it=%d" % g.iter)
        assert_same(g.get("S"), o.S, "S it=%d" % g.iter)
    assert g.eig_epochs >= 5


def test_checkpoint_with_a_pending_per_walker_factorization(mods):
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 20, 3, 5, 30
    kw = dict(weights=(20, 20, 0), cov_update=cu, burn=1000, tskip=10, seed=29, cov_mode="per_walker", eig_mode="ql", eig_lag=2)
    a = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    a.init_state(np.zeros(d))
    a.run(2 * cu + 10)                                            # one launch behind an epoch: its tables are pending
    st = a.checkpoint()
    assert st["eig_pending"] == 1
    a.run(2 * cu + 5)
    b = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
    b.init_state(np.zeros(d))
    b.restore(st)
    b.run(2 * cu + 5)
    for name in ("X", "lnL", "cov", "Ut", "S", "nacc"):
        assert_same(a.get(name), b.get(name), name)
