"""Parity of the kernel instantiations and sizes that bench.py times (BASELINE configs[1..4]), HIP vs oracle, bit for bit.

Every test asserts through ptmi_last_mh_variant that the instantiation it means to test is the one that launched.
Run on the GPU box: ``python -m pytest tests -m gpu``.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from test_gpu_parity import _compare, _pair, assert_same, mods  # noqa: F401  (mods is a fixture)

pytestmark = pytest.mark.gpu


def _dense(d, seed=0):
    """The dense Gaussian of bench.py --logl dense (BASELINE configs[2]; tests/test_simple.py:27-30 in the reference)."""
    A = np.random.default_rng(seed).standard_normal((d, d))
    return ("dense", np.zeros(d), np.linalg.inv(A @ A.T / d + np.eye(d)))


@pytest.mark.parametrize("cov_mode,nt,W,dense512", [("pooled", 4, 9, True), ("per_walker", 3, 5, False), ("per_walker", 64, 2, False),
                                                     ("pooled", 64, 5, True), ("per_walker", 128, 3, True)])
def test_dense_scam_only_runs_staged_and_matches(mods, cov_mode, nt, W, dense512):
    """Dense likelihood, SCAM-only cycle (PTMCMCSampler.py:605-612, 820-876).  Two instantiations serve it and the variant
    flag tells them apart: ``mh_dense_scam_kernel<26, 512>`` (PTMI_VAR_DENSE_SCAM: one table per 128-chain block -- pooled
    covariance, or a walker's ranks filling whole blocks; the kernel bench.py --logl dense times), here on one block, on
    2.5 blocks (64 x 5 chains) and on three walkers with a block each; and the older staged kernel with the eigenvectors
    read per chain.  per_walker with nt = 3 puts several walkers in one block: each chain must use ITS walker's table
    after the first covariance epoch."""
    orc, _lib, _ = mods
    d = 100
    g, o = _pair(mods, d, nt, W, logl=_dense(d), cov0=np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=25, burn=1000,
                 tskip=10, seed=5, cov_mode=cov_mode)
    g.run(80)
    o.run(80)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_STAGED and not flags & _lib.VAR_FULL and (G, E) == (4, 25)
    assert bool(flags & _lib.VAR_DENSE_SCAM) == dense512
    _compare(g, o, "dense scam %s " % cov_mode)
    assert_same(g.get("Ut"), o.Ut, "Ut")
    if cov_mode == "per_walker":
        assert not np.array_equal(o.Ut[0], o.Ut[1])            # the walkers' tables did diverge


@pytest.mark.parametrize("cov_mode,nt,W,weights", [("pooled", 4, 7, (20, 20, 0)), ("pooled", 3, 6, (20, 20, 20)),
                                                    ("per_walker", 64, 2, (20, 20, 20)), ("pooled", 5, 4, (0, 20, 0))])
def test_dense_with_am_runs_on_the_matrix_cores_and_matches(mods, cov_mode, nt, W, weights):
    """BASELINE configs[2] as benchmarked: dense likelihood AND AM proposal as f64-MFMA table products
    (mh_steps_kernel<4,26,1,true,true,false>), PTMCMCSampler.py:605-612, 879-933."""
    orc, _lib, _ = mods
    d = 100
    g, o = _pair(mods, d, nt, W, logl=_dense(d), cov0=np.eye(d) * 0.01, weights=weights, cov_update=30, burn=60,
                 tskip=10, seed=11, cov_mode=cov_mode)
    g.run(130)
    o.run(130)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_STAGED and flags & _lib.VAR_FULL and (G, E) == (4, 25)
    _compare(g, o, "dense am %s " % cov_mode)
    assert_same(g.get("cov"), o.cov, "cov")
    assert_same(g.get("Ut"), o.Ut, "Ut")
    assert o.jstat[..., 1, 1].sum() > 0                        # AM proposals were accepted


@pytest.mark.parametrize("pc", [True, False])
def test_iso_default_mix_pooled_runs_staged(mods, pc, monkeypatch):
    """The default-mix kernels of bench.py --mix default (iso likelihood: the eigenvector table in LDS): mh_pc_kernel (stepper and
    AM-producer waves paired per SIMD, persistent blocks: what the bench times) and, with PTMI_NO_PC=1, the one-wave kernel with
    its AM queue."""
    orc, _lib, _ = mods
    if not pc:
        monkeypatch.setenv("PTMI_NO_PC", "1")
    g, o = _pair(mods, 100, 64, 3, cov0=np.eye(100) * 0.01, weights=(20, 20, 20), cov_update=40, burn=80, tskip=20,
                 seed=3, cov_mode="pooled")
    g.run(170)
    o.run(170)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_STAGED and flags & _lib.VAR_FULL and flags & _lib.VAR_LDS_UT
    if pc:
        assert flags & _lib.VAR_PC and flags & _lib.VAR_PERSISTENT
    else:
        assert flags & _lib.VAR_AMQ and not flags & _lib.VAR_PC    # per-chain picks, a third of them AM: increments through the queue
    assert flags & _lib.VAR_LDS_DRAWT                          # the draws' tables sit behind the queue in LDS
    _compare(g, o, "mix ")
    assert o.jstat[..., 2, 0].sum() > 0


@pytest.mark.parametrize("weights,nt,W,n,queued", [((20, 20, 0), 64, 2, 131, True), ((30, 5, 0), 5, 7, 97, True), ((5, 1, 20), 3, 5, 150, True),
                                                   ((0, 20, 0), 4, 4, 60, False), ((1, 30, 0), 4, 4, 60, False)])
def test_am_queue_matches_in_place_am(mods, weights, nt, W, n, queued, monkeypatch):
    """The AM queue of the one-wave staged full kernel (PTMI_NO_PC=1; increments computed 16 events at a time, four to seven steps
    ahead, through a ring in LDS) against the oracle's AM (PTMCMCSampler.py:879-933): launches of odd lengths, sparse and dense AM
    picks, a wave with dead lanes, DE switching on in between; AM-heavy cycles stay in place."""
    orc, _lib, _ = mods
    monkeypatch.setenv("PTMI_NO_PC", "1")
    d = 100
    g, o = _pair(mods, d, nt, W, cov0=np.eye(d) * 0.01, weights=weights, cov_update=20, burn=40, tskip=7, seed=17, cov_mode="pooled")
    for m in (n, 3, 1, 46):                                    # the queue restarts with every launch, whatever its length
        g.run(m)
        o.run(m)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_STAGED and flags & _lib.VAR_FULL and bool(flags & _lib.VAR_AMQ) == queued
    _compare(g, o, "am queue %r " % (weights,))
    assert o.jstat[..., 1, 0].sum() > 0


@pytest.mark.parametrize("weights,nt,W,n,cov_mode,pick", [((20, 20, 0), 64, 2, 131, "pooled", "chain"), ((30, 5, 0), 5, 7, 97, "pooled", "chain"),
                                                          ((5, 1, 20), 3, 5, 150, "pooled", "chain"), ((0, 20, 0), 4, 4, 60, "pooled", "chain"),
                                                          ((1, 30, 0), 4, 4, 60, "pooled", "chain"), ((20, 20, 20), 64, 3, 120, "per_walker", "chain"),
                                                          ((20, 20, 20), 64, 2, 120, "pooled", "walker"), ((20, 20, 20), 7, 11, 90, "pooled", "walker")])
def test_producer_consumer_kernel_matches_the_oracle(mods, weights, nt, W, n, cov_mode, pick):
    """mh_pc_kernel (stepper + AM-producer wave pairs, the ring handed over through two LDS counters) against the oracle's AM
    (PTMCMCSampler.py:879-933): launches of odd lengths (the event numbers restart with every launch, passes do not straddle
    units), sparse and AM-only cycles, waves with dead lanes, DE switching on in between, a block per walker (per-walker tables: not
    persistent), one pick per walker."""
    orc, _lib, _ = mods
    d = 100
    g, o = _pair(mods, d, nt, W, cov0=np.eye(d) * 0.01, weights=weights, cov_update=20, burn=40, tskip=7, seed=17, cov_mode=cov_mode,
                 pick_mode=pick)
    for m in (n, 3, 1, 46):
        g.run(m)
        o.run(m)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_PC and flags & _lib.VAR_FULL and bool(flags & _lib.VAR_PERSISTENT) == (cov_mode == "pooled")
    _compare(g, o, "pc %r %s %s " % (weights, cov_mode, pick))
    assert o.jstat[..., 1, 0].sum() > 0


@pytest.mark.parametrize("kind", ["scam", "mix", "dense"])
def test_box_prior_from_lds_table(mods, kind):
    """Box prior (-inf outside [pmin, pmax], the reference's usual lnpriorfn) in the kernels bench.py --prior box times: the
    bounds table in LDS (SCAM-only table kernel, staged full kernel, dense 512-thread kernel), walls close enough to be hit."""
    orc, _lib, _ = mods
    d, nt, W = 100, 64, 2
    rs = np.random.RandomState(4)
    lo, hi = -0.25 - rs.rand(d) * 0.1, 0.2 + rs.rand(d) * 0.1      # per-parameter bounds
    kw = dict(cov0=np.eye(d) * 0.01, cov_update=30, burn=60, tskip=10, seed=23, cov_mode="pooled", logp=("box", lo, hi),
              p0=rs.uniform(-0.05, 0.05, (W, nt, d)))
    if kind == "scam":
        kw.update(weights=(20, 0, 0))
    elif kind == "mix":
        kw.update(weights=(20, 20, 20))
    else:
        kw.update(weights=(20, 0, 0), logl=_dense(d))
    g, o = _pair(mods, d, nt, W, **kw)
    g.run(140)
    o.run(140)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_LDS_BOX and (G, E) == (4, 25)
    if kind == "scam":
        assert flags & _lib.VAR_LDS_UT and not flags & _lib.VAR_STAGED
    if kind == "dense":
        assert flags & _lib.VAR_DENSE_SCAM              # the box-prior instantiation of the 512-thread kernel
    _compare(g, o, "box %s " % kind)
    rej = o.jstat[..., 0, 0].sum() - o.jstat[..., 0, 1].sum()
    assert np.isinf(o.lp).sum() == 0 and rej > 0                # proposals did leave the box and were refused


class _Subset(object):
    """A few walkers of a big pooled run on the oracle.  Pooled tables depend on every walker, so the epochs are driven
    from outside: the covariance / DE inputs are the device's AM buffers, run through the ORACLE's pooled statistics,
    factorization and DE update; the chains then continue on the oracle with those tables."""

    def __init__(self, orc, walkers, d, nt, W, cov0, **kw):
        self.orc, self.walkers, self.d, self.W = orc, walkers, d, W
        self.subs = []
        for w0 in walkers:
            o = orc.OracleEngine(d, nt, 1, cov0, cov_mode="pooled", walker0=w0, **kw)
            o._epochs = lambda it: None
            self.subs.append(o)
        self.cu, self.burn = kw["cov_update"], kw["burn"]
        self.mu, self.M2 = np.zeros(d), np.zeros((d, d))
        self.DE = np.zeros((self.burn, d))

    def epoch(self, AM, it_done, flags=None, apply=True):
        """PTMCMCSampler.py:545-585 for iteration it_done + 1, from all walkers' AM rows (flags: the device's AM row flags of an
        engine in am_mode "rle": the statistics then weight every stored row by its run length).  apply=False (eig_lag): the new
        covariance is kept, its factorization takes effect when apply_table() is called."""
        orc, d, W = self.orc, self.d, self.W
        if it_done % self.cu == 0:
            if flags is not None:
                cov_o = orc.pool_update_rle(AM, flags, self.mu, self.M2, it_done)
            else:
                cov_o = orc.pool_update(AM, self.mu, self.M2, it_done)
            for o in self.subs:
                o.cov[0] = cov_o
                if apply:
                    o._svd(0)
        if it_done % self.burn == 0:
            AMc = np.ascontiguousarray(AM)
            orc.lib().orc_de_update_pooled(d, self.burn, self.cu, W, orc._p(self.DE), orc._p(AMc))
            for o in self.subs:
                o.DE[0] = self.DE
        if it_done == self.burn:
            for o in self.subs:
                o.cfg.de_on = 1


    def apply_table(self):
        for o in self.subs:
            o._svd(0)


def _check_subset(g, sub, what):
    X, lnL, so = g.get("X"), g.get("lnL"), g.get("slot_of")
    nsw, nacc, js = g.get("nswap"), g.get("nacc"), g.get("jstat")
    for w0, o in zip(sub.walkers, sub.subs):
        assert_same(X[w0], o.X[0], "%s walker %d X" % (what, w0))
        assert_same(lnL[w0], o.lnL[0], "%s walker %d lnL" % (what, w0))
        assert_same(so[w0], o.slot_of[0], "%s walker %d slot_of" % (what, w0))
        assert_same(nsw[w0], o.nswap[0], "%s walker %d nswap" % (what, w0))
        assert_same(nacc[w0], o.nacc[0], "%s walker %d nacc" % (what, w0))
        assert_same(js[w0], o.jstat[0], "%s walker %d jstat" % (what, w0))


def test_full_size_default_mix_with_covariance_and_de_epochs(mods):
    """BASELINE configs[1] size (64 temps x 4096 walkers x 100-d), default SCAM/AM/DE mix, pooled covariance with
    covUpdate = 100 and burn = 200, 400 iterations: three covariance epochs (matrix-core Welford over 4096 walkers +
    two-level pooling), a DE epoch and DE activation, four swap epochs.  Device covariance / eigenvectors / DE ring
    equal the oracle's on the same AM rows, and three walkers' chains match the oracle bit for bit throughout."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 20, 20), cov_update=100, burn=200, tskip=100, seed=1234)
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", **kw)
    g.init_state(np.zeros(d))
    sub = _Subset(orc, (0, 1777, 4095), d, nt, W, cov0, **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
    cu = kw["cov_update"]
    seen = 0
    for k in range(4):
        if k > 0:
            g.sync()
            sub.epoch(g.get("AM"), k * cu, g.get("AMflag") if g.am_rle else None)
        g.run(cu)                                   # starts with the device's own epoch
        for o in sub.subs:
            o.run(cu)
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_STAGED and flags & _lib.VAR_FULL and flags & _lib.VAR_LDS_UT and (G, E) == (4, 25)
        if k > 0:
            assert_same(g.get("cov")[0], sub.subs[0].cov[0], "pooled cov after epoch %d" % k)
            assert_same(g.get("Ut")[0], sub.subs[0].Ut[0], "Ut after epoch %d" % k)
            assert_same(g.get("S")[0], sub.subs[0].S[0], "S after epoch %d" % k)
        if k * cu >= kw["burn"]:
            ring = g.get("DE")[0]
            assert_same(np.roll(ring, -g.de_head, axis=0), sub.DE, "DE history")
            seen += 1
        _check_subset(g, sub, "segment %d" % k)
    assert seen == 2 and g.de_on
    js = g.get("jstat").astype(np.int64)
    assert (js[..., :3, 0].sum(-1) == 400).all() and js[..., 2, 0].sum() > 0 and js[..., 1, 1].sum() > 0
    so = g.get("slot_of")
    assert (np.sort(so, axis=1) == np.arange(nt)).all()
    assert g.get("nswap").sum() > 0 and g.swap_proposed == 4


def test_full_size_dense_scam_through_pooled_covariance_epochs(mods):
    """BASELINE configs[2] at full size (100-d dense Gaussian, 64 temps x 4096 walkers = 2048 blocks of
    mh_dense_scam_kernel<26, 512>), SCAM cycle, pooled covariance with covUpdate = 100: 300 iterations = two pooled
    covariance epochs (matrix-core Welford over 4096 walkers, two-level pooling, factorization) and three swap epochs.
    Pooled cov / Ut / S equal the oracle's on the same AM rows and three walkers' chains (first, middle, last block) match
    the oracle bit for bit throughout (PTMCMCSampler.py:605-612, 820-876, 769-803; tests/test_simple.py:27-30 is the
    reference's own dense Gaussian)."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 0, 0), cov_update=100, burn=10000, tskip=100, seed=77, logl=_dense(d))
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", **kw)
    g.init_state(np.zeros(d))
    sub = _Subset(orc, (0, 2049, 4095), d, nt, W, cov0, **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
    cu = kw["cov_update"]
    for k in range(3):
        if k > 0:
            g.sync()
            sub.epoch(g.get("AM"), k * cu, g.get("AMflag") if g.am_rle else None)
        g.run(cu)
        for o in sub.subs:
            o.run(cu)
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_DENSE_SCAM and flags & _lib.VAR_STAGED and not flags & _lib.VAR_FULL and (G, E) == (4, 25)
        if k > 0:
            assert_same(g.get("cov")[0], sub.subs[0].cov[0], "pooled cov after epoch %d" % k)
            assert_same(g.get("Ut")[0], sub.subs[0].Ut[0], "Ut after epoch %d" % k)
            assert_same(g.get("S")[0], sub.subs[0].S[0], "S after epoch %d" % k)
        _check_subset(g, sub, "dense segment %d" % k)
    # size-independent properties over the whole batch: lnL is the quadratic form of the held row, the tables are permutations
    X, lnL = g.get("X"), g.get("lnL")
    P = kw["logl"][2]
    assert np.allclose(lnL, -0.5 * np.einsum("wti,ij,wtj->wt", X, P, X), rtol=1e-11, atol=1e-11)
    so = g.get("slot_of")
    assert (np.sort(so, axis=1) == np.arange(nt)).all()
    js = g.get("jstat").astype(np.int64)
    assert (js[..., 0, 0] == 300).all() and g.get("nswap").sum() > 0 and g.swap_proposed == 3


def test_full_size_scam_persistent_kernel_on_adapted_tables(mods):
    """BASELINE configs[1] at full size and AS BENCHMARKED -- mh_steps_kernel<4,25,iso,...,ULDS,512> (persistent blocks over one
    LDS copy of the pooled eigenvector table, cold-first walk) -- on ADAPTED, dense tables: covUpdate = 100, 300 iterations = two
    pooled covariance epochs and three swap epochs (with covUpdate = 1000 a short run only ever sees the start's unit vectors,
    24 of a lane's 25 increments zero).  Pooled cov / Ut / S equal the oracle's on the same AM rows, and three walkers' chains
    (first, middle, last) match the oracle bit for bit throughout (PTMCMCSampler.py:820-876, 605-622, 631-697, 769-803)."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 0, 0), cov_update=100, burn=10000, tskip=100, seed=4321)
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", **kw)
    g.init_state(np.zeros(d))
    sub = _Subset(orc, (0, 2047, 4095), d, nt, W, cov0, **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
    cu = kw["cov_update"]
    for k in range(3):
        if k > 0:
            g.sync()
            sub.epoch(g.get("AM"), k * cu, g.get("AMflag") if g.am_rle else None)
        g.run(cu)
        for o in sub.subs:
            o.run(cu)
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_PERSISTENT and flags & _lib.VAR_LDS_UT and not flags & _lib.VAR_FULL and (G, E) == (4, 25)
        if k > 0:
            assert_same(g.get("cov")[0], sub.subs[0].cov[0], "pooled cov after epoch %d" % k)
            assert_same(g.get("Ut")[0], sub.subs[0].Ut[0], "Ut after epoch %d" % k)
            assert_same(g.get("S")[0], sub.subs[0].S[0], "S after epoch %d" % k)
            Ut = g.get("Ut")[0, 0]
            assert (np.abs(Ut) > 1e-6).mean() > 0.9              # the adapted table is dense: every increment of a lane is live
        _check_subset(g, sub, "scam segment %d" % k)
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * (X ** 2).sum(-1), rtol=1e-12, atol=1e-12)
    so = g.get("slot_of")
    assert (np.sort(so, axis=1) == np.arange(nt)).all()
    js = g.get("jstat").astype(np.int64)
    assert (js[..., 0, 0] == 300).all() and g.get("nswap").sum() > 0 and g.swap_proposed == 3


def test_full_size_scam_as_benchmarked_late_table_and_run_length_rows(mods):
    """BASELINE configs[1] at full size in the configuration bench.py times (PTMCMCSampler.py:545-560, 769-803): the persistent SCAM
    kernel with am_mode = rle (a step stores its row only when it was accepted; run-length-weighted pooled statistics) AND
    eig_lag = 1 (the table of a covariance epoch takes effect one launch late, the host factorizing meanwhile).  covUpdate = 100,
    400 iterations = three epochs, each table in force one launch after its epoch; three walkers bit for bit throughout."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 0, 0), cov_update=100, burn=10000, tskip=100, seed=97)
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", am_mode="rle", eig_lag=1, **kw)
    assert g.am_rle and g.eig_lag == 1
    g.init_state(np.zeros(d))
    sub = _Subset(orc, (1, 2048, 4094), d, nt, W, cov0, am_mode="rle", **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
    cu = kw["cov_update"]
    tables = [g.get("Ut").copy()]
    for k in range(4):
        if k > 0:
            g.sync()
            sub.epoch(g.get("AM"), k * cu, g.get("AMflag"), apply=False)       # the statistics now, the table after the next launch
        g.run(cu)
        for o in sub.subs:
            o.run(cu)
        if k > 0:
            sub.apply_table()
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_PERSISTENT and flags & _lib.VAR_LDS_UT and not flags & _lib.VAR_FULL and (G, E) == (4, 25)
        _check_subset(g, sub, "as benchmarked, segment %d" % k)              # segment k ran with the table of epoch k - 1
        if k > 0:
            assert_same(g.get("cov")[0], sub.subs[0].cov[0], "pooled cov after epoch %d" % k)
            assert_same(g.get("Ut")[0], sub.subs[0].Ut[0], "Ut in force after segment %d" % k)
            assert_same(g.get("S")[0], sub.subs[0].S[0], "S after segment %d" % k)
        tables.append(g.get("Ut").copy())
    assert all(not np.array_equal(a, b) for a, b in zip(tables[1:-1], tables[2:]))       # every epoch changed the table
    fl = g.get("AMflag")
    assert 0 < ((fl & 3) == 0).mean() < 1                           # some rows were not stored
    assert g.eig_epochs == 3


def test_config4_slice_1000d_64_temps(mods):
    """BASELINE configs[3], one GPU's share: 1000-d isotropic Gaussian, 64 ranks x 512 walkers, default mix (64 lanes per
    chain); invariants on the batch and bit parity on two walkers, through a swap epoch."""
    orc, _lib, PTEngine = mods
    d, nt, W, n = 1000, 64, 512, 100
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 20, 20), cov_update=1000, burn=10000, tskip=50, seed=99)
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", **kw)
    g.init_state(np.zeros(d))
    g.run(n)
    g.sync()
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_FULL and G == 64
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * (X ** 2).sum(-1), rtol=1e-12)
    js = g.get("jstat").astype(np.int64)
    assert (js[..., :2, 0].sum(-1) == n).all() and js[..., 1, 0].sum() > 0
    sub = _Subset(orc, (3, 511), d, nt, W, cov0, **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
        o.run(n)
    _check_subset(g, sub, "config 4")
    assert g.get("nswap").sum() > 0


def test_config4_share_scam_cycle_as_benchmarked(mods):
    """BASELINE configs[3], one GPU's share AS bench.py --ndim 1000 times it: 64 ranks x 512 walkers x 1000-d, SCAM cycle, pooled
    covariance -- the 64-lane kernel with wide draw batches over the library's padded table copy (PTMI_VAR_UTPAD;
    PTMCMCSampler.py:820-876, 605-622) -- through a pooled covariance epoch (covUpdate = 100: the second segment runs on an adapted,
    dense 1000 x 1000 table; host factorization, so that the table is the oracle's bit for bit) and two swap epochs."""
    orc, _lib, PTEngine = mods
    d, nt, W = 1000, 64, 512
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 0, 0), cov_update=100, burn=10000, tskip=100, seed=5)
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", **kw)
    g.init_state(np.zeros(d))
    sub = _Subset(orc, (0, 300, 511), d, nt, W, cov0, am_mode="rle" if g.am_rle else "rows", **kw)
    for o in sub.subs:
        o.init_state(np.zeros(d))
    for k in range(2):
        if k > 0:
            g.sync()
            sub.epoch(g.get("AM"), k * 100, g.get("AMflag") if g.am_rle else None)
        g.run(100)
        for o in sub.subs:
            o.run(100)
        flags, G, E = g.last_variant()
        assert flags & _lib.VAR_UTPAD and not flags & _lib.VAR_FULL and (G, E) == (64, 16)
        if k > 0:
            assert_same(g.get("Ut")[0], sub.subs[0].Ut[0], "Ut after the epoch")
            assert (np.abs(g.get("Ut")[0, 0]) > 1e-9).mean() > 0.9
        _check_subset(g, sub, "config 4 share, segment %d" % k)
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * (X ** 2).sum(-1), rtol=1e-12, atol=1e-12)
    assert g.get("nswap").sum() > 0 and g.swap_proposed == 2


def test_config5_slice_curved_nuts_16_temps(mods):
    """BASELINE configs[4], one GPU's share: 20-d curved likelihood, 16 ranks x 4096 walkers, SCAM + DE + NUTS cycle
    (DE still waiting for burn), box prior; bit parity on three walkers incl. their NUTS step-size state."""
    orc, _lib, PTEngine = mods
    d, nt, W, n = 20, 16, 4096, 120
    cov0 = np.eye(d)
    p0 = np.array([-0.1, -0.5] * (d // 2))
    kw = dict(logl=("curved",), logp=("box", np.full(d, -10.0), np.full(d, 10.0)), weights=(10, 0, 10), grad_weights=(10, 0),
              cov_update=1000, burn=10000, tskip=50, seed=7)
    g = PTEngine(d, nt, W, cov0, cov_mode="pooled", **kw)
    g.init_state(p0)
    g.run(n)
    g.sync()
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_GRADJUMP
    sub = _Subset(orc, (0, 2048, 4095), d, nt, W, cov0, **kw)
    for o in sub.subs:
        o.init_state(p0)
        o.run(n)
    _check_subset(g, sub, "config 5")
    gj = g.get("gj")
    for w0, o in zip(sub.walkers, sub.subs):
        assert_same(gj[w0], o.gj[0], "walker %d NUTS state" % w0)
    assert g.get("jstat").astype(np.int64)[..., 3, 0].sum() > 0


@pytest.mark.parametrize("nt,W,cov_mode,logl", [(64, 3, "pooled", "iso"), (16, 5, "per_walker", "iso"), (3, 7, "pooled", "iso"),
                                                (64, 2, "pooled", "dense"), (5, 3, "per_walker", "dense")])
def test_walker_pick_mode_matches_oracle_and_is_uniform(mods, nt, W, cov_mode, logl):
    """pick_mode="walker" (include/ptmi.h): one cycle draw per walker and iteration, from the stream of its rank 0,
    fixes the proposal TYPE of all its ranks; everything else stays per chain.  HIP == oracle bit for bit, and the
    per-type proposal counts are identical across the ranks of a walker."""
    orc, _lib, _ = mods
    d = 100
    g, o = _pair(mods, d, nt, W, logl=_dense(d) if logl == "dense" else ("iso",), cov0=np.eye(d) * 0.01, weights=(20, 20, 20),
                 cov_update=30, burn=60, tskip=10, seed=17, cov_mode=cov_mode, pick_mode="walker")
    g.run(150)
    o.run(150)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_UNIFORM and flags & _lib.VAR_FULL
    _compare(g, o, "walker pick ")
    js = g.get("jstat").astype(np.int64)[..., :3, 0]                      # [W][nt][type] proposals
    assert (js == js[:, :1]).all() and (js.sum(-1) == 150).all()
    assert (js[:, 0] > 0).all()                                           # every type was drawn
    if W > 1:
        assert not (js[0, 0] == js[1:, 0]).all()                          # walkers draw independently


def test_walker_pick_mode_with_gradient_jumps(mods):
    g, o = _pair(mods, 20, 4, 5, logl=("curved",), logp=("box", -10 * np.ones(20), 10 * np.ones(20)), cov0=np.eye(20),
                 p0=np.tile(np.array([-0.1, -0.5] * 10), (5, 4, 1)), weights=(10, 0, 10), grad_weights=(10, 5),
                 cov_update=50, burn=60, tskip=10, seed=23, pick_mode="walker")
    g.run(130)
    o.run(130)
    _compare(g, o, "walker pick + NUTS ")
    assert_same(g.get("gj"), o.gj, "gj")
    js = g.get("jstat").astype(np.int64)[..., 0]
    assert (js == js[:, :1]).all() and js[..., 3].sum() > 0 and js[..., 4].sum() > 0


def test_full_size_stationary_distribution_at_every_temperature(mods):
    """BASELINE configs[1] as benchmarked (64 temps x 4096 walkers x 100-d, SCAM cycle, pooled covariance, swaps every 100):
    after 40 000 iterations from p0 = 0 the batch samples what it should.  Rank t of an isotropic Gaussian at temperature T_t
    holds x ~ N(0, T_t I), so <lnL> = -d T_t / 2 with standard deviation sqrt(d / 2) T_t -- checked across the 4096 independent
    walkers for every rank up to T = 100 (above, the reference does not scale its jumps with sqrt(T), PTMCMCSampler.py:861-862,
    and the hot chains need far longer) -- and the cold chains have mean 0 and variance 1 in every parameter.  A property of
    the whole path at full size: proposals, accept test, swaps and adaptation together (the oracle cannot run this size)."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=11, cov_mode="pooled")
    g.init_state(np.zeros(d))
    g.run(40000)
    g.sync()
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_LDS_UT and flags & _lib.VAR_PERSISTENT and (G, E) == (4, 25)
    lnL, T = g.by_temp("lnL"), g.ladder
    warm = T <= 100.0
    assert warm.sum() >= 30
    z = (lnL.mean(0) + 0.5 * d * T) / (lnL.std(0) / np.sqrt(W))
    assert np.abs(z[warm]).max() < 5.0, z[warm]
    assert np.allclose(lnL.std(0)[warm], np.sqrt(d / 2.0) * T[warm], rtol=0.06)
    X = g.by_temp("X")[:, 0]                                  # the 4096 cold states
    assert np.abs(X.mean(0)).max() < 5.0 / np.sqrt(W)
    assert np.abs(X.var(0) - 1.0).max() < 5.0 * np.sqrt(2.0 / W)
    S = g.get("S")[0, 0]
    assert 0.9 < S.min() and S.max() < 1.1                    # the adapted covariance found the unit target
    acc = g.get("nswap").astype(np.float64)[:, :nt - 1].mean(0) / g.swap_proposed
    assert 0.2 < acc[:30].min() and acc[:30].max() < 0.8      # the ladder's design acceptance (tstep = 1 + sqrt(2 / d), :711)


def test_full_size_dense_target_covariance(mods):
    """BASELINE configs[2] as benchmarked (100-d dense Gaussian, 64 x 4096 chains, the 512-thread matrix-core kernel with the
    half-matrix quadratic form): after 30 000 iterations the 4096 cold states have the target's covariance P^-1, entry by entry
    within five standard errors, and <lnL> = -d T / 2 holds down the warm part of the ladder."""
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    A = np.random.default_rng(0).standard_normal((d, d))
    Ctrue = A @ A.T / d + np.eye(d)
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=5, cov_mode="pooled",
                 logl=("dense", np.zeros(d), np.linalg.inv(Ctrue)))
    g.init_state(np.zeros(d))
    g.run(30000)
    g.sync()
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_DENSE_SCAM and (G, E) == (4, 25)
    X = g.by_temp("X")[:, 0]
    Chat = X.T @ X / W
    se = np.sqrt((np.outer(np.diag(Ctrue), np.diag(Ctrue)) + Ctrue ** 2) / W)
    assert np.abs((Chat - Ctrue) / se).max() < 5.5, np.abs((Chat - Ctrue) / se).max()
    assert np.abs(X.mean(0) / np.sqrt(np.diag(Ctrue) / W)).max() < 5.0
    lnL, T = g.by_temp("lnL"), g.ladder
    warm = T <= 100.0
    z = (lnL.mean(0) + 0.5 * d * T) / (lnL.std(0) / np.sqrt(W))
    assert np.abs(z[warm]).max() < 5.0, z[warm]
    # the pooled adaptive covariance (cumulative since p0 = 0, transient included) is on its way to the target's
    assert np.linalg.norm(g.get("cov")[0] - Ctrue) < 0.3 * np.linalg.norm(Ctrue)


@pytest.mark.parametrize("d,nt,W,cov_mode", [(50, 64, 2, "pooled"), (80, 5, 7, "pooled"), (104, 8, 3, "pooled"), (50, 64, 2, "per_walker")])
def test_dense_likelihood_producer_consumer_kernel_at_other_shapes(mods, d, nt, W, cov_mode):
    """The dense Gaussian's default mix below 100-d (shapes (4, 14), (4, 20), (4, 26)): the producer / consumer kernel with the likelihood's
    table in LDS and the eigenvectors read from global memory, as at 100-d (round 5: with both tables in LDS these shapes fell back
    to the one-wave kernel: 11.8 / 18.8 ms per 100 steps at 50 / 80-d against 7.5 / 11.6 now).  HIP == oracle, the variant asserted."""
    orc, _lib, _ = mods
    rs = np.random.RandomState(d)
    A = rs.randn(d, d)
    logl = ("dense", rs.randn(d) * 0.1, np.linalg.inv(A @ A.T / d + 0.5 * np.eye(d)))
    g, o = _pair(mods, d, nt, W, cov0=np.eye(d) * 0.01, weights=(20, 20, 20), cov_update=20, burn=40, tskip=7, seed=23, cov_mode=cov_mode, logl=logl)
    for m in (97, 3, 1, 46):
        g.run(m)
        o.run(m)
    flags, G, E = g.last_variant()
    assert flags & _lib.VAR_PC and flags & _lib.VAR_FULL and not flags & _lib.VAR_LDS_UT
    _compare(g, o, "dense pc d=%d %s " % (d, cov_mode))
    assert o.jstat[..., :3, 0].sum(axis=(0, 1)).min() > 0
