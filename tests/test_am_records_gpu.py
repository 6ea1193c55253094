"""AM records (include/ptmi.h ``ptmi_buffers.AMrec``): the rank-0 chain's samples (updateChains' buffer,
PTMCMCSampler.py:327-328) kept as 16-byte step records between KEY rows.  Everything a reader sees -- the rows
(``ptmi_am_expand``), the pooled covariance the statistics kernel rebuilds them for, the chains -- is bit for bit what a run
that stores every row sees, and what the oracle computes.

Run on the GPU box: ``python -m pytest tests -m gpu``.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from test_gpu_parity import _compare, _pair, assert_same, mods  # noqa: F401  (mods is a fixture)

pytestmark = pytest.mark.gpu


def _dense(d, seed=0):
    A = np.random.default_rng(seed).standard_normal((d, d))
    return ("dense", np.zeros(d), np.linalg.inv(A @ A.T / d + np.eye(d)))


CASES = [
    # d, nt, W, cov_update, tskip, extra
    (100, 64, 3, 40, 20, {}),                                   # the persistent kernel of the exact shape, swaps, three epochs
    (100, 5, 9, 25, 7, {}),                                     # ring shorter than two chunks, swap period that does not divide it
    (100, 1, 6, 30, 0, {}),                                     # one temperature: no swap ever writes a KEY row
    (100, 64, 2, 50, 25, {"logl": "dense"}),                    # mh_dense_scam_kernel
    (100, 8, 4, 33, 11, {"box": True}),                         # box prior: rejected out-of-box proposals are records too
    (37, 6, 5, 20, 10, {}),                                     # a 4-lane shape with padding slots
    (7, 3, 4, 16, 4, {}),
    (130, 4, 3, 24, 8, {}),                                     # 16 lanes per chain, two macro tiles in the statistics
    (500, 2, 2, 16, 8, {}),                                     # 64 lanes per chain
]


@pytest.mark.parametrize("d,nt,W,cu,tskip,extra", CASES)
def test_records_equal_rows_and_the_oracle(mods, d, nt, W, cu, tskip, extra):
    orc, _lib, PTEngine = mods
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=tskip, seed=31, cov_mode="pooled", cov0=np.eye(d) * 0.01)
    if extra.get("logl") == "dense":
        kw["logl"] = _dense(d)
    if extra.get("box"):
        rs = np.random.RandomState(4)
        kw["logp"] = ("box", -0.25 - rs.rand(d) * 0.1, 0.2 + rs.rand(d) * 0.1)
        kw["p0"] = rs.uniform(-0.05, 0.05, (W, nt, d))
    g, o = _pair(mods, d, nt, W, am_mode="records", **kw)
    r, _ = _pair(mods, d, nt, W, am_mode="rows", **kw)
    assert g.am_records and not r.am_records
    total = 0
    for n in (cu + 3, 1, 2 * cu - 5, 17, cu):                   # launches of odd lengths, epochs inside
        g.run(n)
        r.run(n)
        o.run(n)
        total += n
        _compare(g, o, "records d=%d it=%d " % (d, total))      # get("AM") expands the current covariance period
        for name in ("X", "lnL", "cov", "Ut", "S", "mu", "M2", "nacc", "slot_of"):
            assert_same(g.get(name), r.get(name), "records vs rows %s d=%d it=%d" % (name, d, total))
        lo, hi = g.am_period()
        rows = np.arange(lo, hi + 1) % cu
        assert_same(g.get("AM")[:, rows], r.get("AM")[:, rows], "records vs rows AM d=%d it=%d" % (d, total))
    assert_same(g.get("cov"), o.cov, "cov")
    rec = g.t["AMrec"].cpu().numpy()
    key = (rec[..., 1] >> 33) & 1
    assert 0 < key.mean() < 0.6                                 # most rows are records
    if extra.get("box"):
        assert o.jstat[..., 0, 0].sum() > o.jstat[..., 0, 1].sum()


def test_expand_of_a_range_leaves_the_other_rows_alone(mods):
    orc, _lib, PTEngine = mods
    d, nt, W, cu = 100, 4, 6, 50
    kw = dict(weights=(20, 0, 0), cov_update=cu, burn=1000, tskip=10, seed=5, cov_mode="pooled", cov0=np.eye(d) * 0.01)
    g, o = _pair(mods, d, nt, W, am_mode="records", **kw)
    g.run(130)
    o.run(130)
    g.sync()
    before = g.am_params(g.t["AM"].cpu().numpy())               # KEY rows valid, the others stale
    g.am_expand(2, 3, 111, 125)                                 # walkers 2..4, iterations 111..125
    after = g.am_params(g.t["AM"].cpu().numpy())
    rows = np.arange(111, 126) % cu
    assert_same(after[2:5][:, rows], o.AM[2:5][:, rows], "expanded range")
    mask = np.ones(after.shape[:2], bool)
    mask[2:5, rows] = False
    assert_same(after[mask], before[mask], "rows outside the range")
    assert not np.array_equal(before[2:5][:, rows], o.AM[2:5][:, rows])      # they did need rebuilding


def test_records_mode_is_refused_where_rows_are_needed(mods):
    orc, _lib, PTEngine = mods
    with pytest.raises(ValueError):
        PTEngine(10, 2, 2, np.eye(10), weights=(20, 20, 0), cov_mode="pooled", am_mode="records")
    with pytest.raises(ValueError):
        PTEngine(10, 2, 2, np.eye(10), weights=(20, 0, 0), cov_mode="per_walker", am_mode="records")
    g = PTEngine(10, 2, 2, np.eye(10), weights=(20, 20, 20), cov_mode="pooled")      # auto: rows
    assert not g.am_records and g.t["AMrec"] is None


def test_checkpoint_of_a_records_run_continues_bit_identically(mods):
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
    rows = np.arange(*a.am_period()) % 40
    assert_same(a.get("AM")[:, rows], b.get("AM")[:, rows], "AM")
