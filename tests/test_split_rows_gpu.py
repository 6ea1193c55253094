"""The split path on contiguous rows (csrc/ptmi_split.hip: ptmi_propose / ptmi_accept / ptmi_accept_propose for likelihoods that
live in a batched device callback, PTMCMCSampler.py:601-622 with :1072-1086 as the boundary).

Three comparisons, all bit for bit:
  * against the ORACLE (orc_mh_steps: the reference's proposals, accept test, swaps and epochs), with the callback returning the
    oracle's own likelihood bits so that the decisions are comparable;
  * against the shape kernels' split path (propose_kernel / accept_kernel of csrc/ptmi_mh.inc.h, PTMI_SPLIT_ROWS=0: what served this
    path until round 5 and still serves cycles with AM entries), every buffer after every call;
  * ptmi_accept_propose (one launch) against ptmi_accept + ptmi_propose (two).

Run on the GPU box: ``python -m pytest tests -m gpu``.  Nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import _compare, _pair, assert_same, mods  # noqa: F401  (mods is a fixture)

pytestmark = pytest.mark.gpu

NAMES = ("X", "lnL", "lp", "temp_of", "slot_of", "nacc", "jstat", "nswap", "Q", "qaux", "AM", "AMaux", "AMflag", "cov", "Ut", "S", "DE")


def _snapshot(g):
    g.sync()
    out = {k: g.t[k].cpu().numpy().copy() for k in NAMES if g.t.get(k) is not None}
    out["Q"] = g.proposals().cpu().numpy().copy()              # the buffer that holds the last proposals (Q or Q2)
    assert not g.t["sloc"].any()                               # outside a segment every state is in X
    return out


def _same(a, b, what):
    assert a.keys() == b.keys()
    for k in a:
        assert_same(a[k], b[k], "%s: %s" % (what, k))


def _callbacks(torch, lo=None, hi=None):
    """An isotropic Gaussian and (optionally) a box prior as BATCHED device callbacks."""
    def logl(X):
        return -0.5 * (X * X).sum(-1)

    if lo is None:
        return logl, None
    lo_t, hi_t = torch.as_tensor(lo, device="cuda"), torch.as_tensor(hi, device="cuda")

    def logp(X):
        inside = ((X >= lo_t) & (X <= hi_t)).all(-1)
        return torch.where(inside, 0.0, -float("inf")).to(torch.float64)

    return logl, logp


CASES = [
    # d, nt, W, weights, extra
    (100, 4, 37, (20, 0, 20), {}),                                  # 148 chains: 2.3 tiles; DE joins after burn
    (100, 64, 3, (20, 0, 0), dict(cov_mode="per_walker")),          # a table per walker
    (101, 3, 5, (20, 0, 20), {}),                                   # odd ndim: 8-byte pieces
    (7, 2, 9, (5, 0, 3), dict(groups=[[0, 2, 4], [1, 3, 5, 6]])),   # parameter groups: the group's table, DE's mask
    (130, 5, 4, (20, 0, 20), dict(pick_mode="walker")),             # 16 lanes per chain in the shape kernels: DE rows in parameter order
    (1000, 3, 2, (20, 0, 20), {}),
    (100, 4, 6, (20, 0, 20), dict(box=True)),                       # a prior that refuses (-inf) in the callback
    (20, 4, 5, (3, 0, 2), dict(w_host=2)),                          # host-served cycle entries: the state handed back unchanged
    (100, 8, 5, (20, 0, 20), dict(cov_mode="pooled", keep_lnl=True)),
    # cycles with AM entries: the increments from the matrix cores ahead of the proposals (ptmi_split_am_prepare)
    (100, 4, 37, (20, 20, 20), {}),
    (100, 64, 2, (20, 20, 0), dict(cov_mode="per_walker")),         # a table -- and a list of picks -- per walker
    (130, 3, 4, (20, 20, 20), dict(cov_mode="pooled")),             # 16 lanes per chain: the Box-Muller pairing (k, k + 16)
    (7, 2, 9, (5, 4, 3), dict(groups=[[0, 2, 4], [1, 3, 5, 6]])),   # a list of picks per parameter group
    (1000, 2, 2, (20, 20, 0), {}),
    (100, 4, 6, (20, 20, 20), dict(small_pieces=True)),             # room for two iterations' increments: several pieces per segment
    (20, 4, 5, (3, 2, 2), dict(w_host=2)),                          # AM entries TOGETHER with host-served ones: the shape kernels serve all three engines
    (1, 2, 3, (20, 0, 20), {}),                                     # one parameter: a row is one 8-byte piece
    (2, 1, 1, (20, 20, 20), {}),                                    # ONE chain: a tile of one row, no ladder
    (104, 3, 70, (20, 20, 20), {}),                                 # the 4-lane shape's largest ndim (DE rows of 104 = 8 x 13 doubles: piece order)
]


@pytest.mark.parametrize("d,nt,W,weights,extra", CASES)
def test_row_kernels_equal_the_shape_kernels_and_one_launch_equals_two(mods, d, nt, W, weights, extra, monkeypatch):
    """Three engines on the same configuration: (a) row kernels, accept + next proposal in one launch (run_callback's default);
    (b) row kernels, two launches per iteration; (c) the shape kernels' split path.  Covariance epochs, DE epochs and activation,
    swaps (whose iterations are accepted by ptmi_accept, the post-swap row written by the swap) all inside; every buffer equal at
    every checkpoint."""
    import torch
    orc, _lib, PTEngine = mods
    extra = dict(extra)
    box = extra.pop("box", False)
    if extra.pop("small_pieces", False):
        monkeypatch.setenv("PTMI_SPLIT_AM_BUDGET_MB", "%.6f" % (2.5 * d * 8 * nt * W / 1048576.0))
    rs = np.random.RandomState(d + nt)
    lo, hi = (-0.4 - 0.1 * rs.rand(d), 0.4 + 0.1 * rs.rand(d)) if box else (None, None)
    logl, logp = _callbacks(torch, lo, hi)
    A = rs.randn(d, d)
    cov0 = (A @ A.T / d + 0.5 * np.eye(d)) * 0.01
    p0 = rs.randn(W, nt, d) * 0.05
    kw = dict(weights=weights, cov_update=20, burn=40, tskip=7, seed=31, split=True, **extra)
    engines = []
    for mode in ("rows fused", "rows two launches", "shape kernels"):
        g = PTEngine(d, nt, W, cov0, **kw)
        g.init_state_callback(p0, logl, logp)
        engines.append((mode, g))
    snaps = {}
    for n in (25, 3, 1, 46, 30):
        for mode, g in engines:
            if mode == "shape kernels":
                monkeypatch.setenv("PTMI_SPLIT_ROWS", "0")
            else:
                monkeypatch.delenv("PTMI_SPLIT_ROWS", raising=False)
            g.run_callback(n, logl, logp, fused=(mode == "rows fused"))
            snaps[mode] = _snapshot(g)
        monkeypatch.delenv("PTMI_SPLIT_ROWS", raising=False)
        it = engines[0][1].iter
        a, b, c = (snaps[m] for m, _ in engines)
        _same(b, c, "two launches vs shape kernels at iteration %d" % it)
        # one launch: qaux[.][2] of the last iteration is the decision in every mode (the segment's last accept is ptmi_accept)
        _same(a, b, "one launch vs two at iteration %d" % it)
    g = engines[0][1]
    js = g.get("jstat").astype(np.int64)
    assert (js[..., 0].sum(-1) + (0 if not extra.get('w_host') else 0) <= g.iter).all()
    if not extra.get('w_host'):
        assert (js[..., 0].sum(-1) == g.iter).all()
    if weights[2]:
        assert js[..., 2, 0].sum() > 0                                                  # DE was proposed after burn
    assert 0 < g.get("nacc").sum() < W * nt * g.iter
    if box:
        assert (js[..., 0].sum(-1) > js[..., 1].sum(-1)).all()
    if nt > 1:
        assert g.get("nswap").sum() > 0
    assert g.eig_epochs >= 4 or d == 1


@pytest.mark.parametrize("d,nt,W,weights,cov_mode", [(100, 4, 5, (20, 0, 20), "per_walker"), (100, 3, 4, (20, 0, 0), "pooled"),
                                                     (37, 5, 3, (20, 0, 20), "pooled"), (6, 2, 3, (1, 0, 1), "per_walker"),
                                                     (100, 4, 5, (20, 20, 20), "per_walker"), (37, 5, 3, (20, 20, 20), "pooled"),
                                                     (300, 2, 3, (5, 20, 5), "pooled")])
def test_row_kernels_against_the_oracle(mods, d, nt, W, weights, cov_mode):
    """The whole callback path on the row kernels against the oracle's run (PTMCMCSampler.py:601-622, 820-876, 936-985, 631-697,
    545-585): the callback hands back the ORACLE's likelihood of every proposal (orc_logl on the host: a torch reduction sums in
    another order), so every decision, state, ring row, covariance and table is comparable bit for bit."""
    import torch
    orc, _lib, PTEngine = mods
    # (am_mode "rows": the split path stores every rank-0 row, and the pooled statistics over stored rows sum in the rows' order)
    g, o = _pair(mods, d, nt, W, weights=weights, cov_update=25, burn=50, tskip=10, seed=8, split=True, cov_mode=cov_mode, am_mode="rows")
    calls = [0]

    def logl(X):
        calls[0] += 1
        q = X.cpu().numpy()
        v = np.array([orc.lib().orc_logl(C.byref(o.cfg), q[i].ctypes.data_as(orc._dp)) for i in range(len(q))])
        return torch.from_numpy(v).to(X.device)

    g.init_state_callback(g.get("X"), logl, None)
    for n in (60, 7, 63):
        g.run_callback(n, logl, None)
        o.run(n)
        _compare(g, o, "rows vs oracle %s it=%d " % (cov_mode, g.iter))
        assert_same(g.get("cov"), o.cov, "cov")
        assert_same(g.get("Ut"), o.Ut, "Ut")
    assert calls[0] == 131 and o.nswap.sum() > 0
    if weights[2]:
        assert o.jstat[..., 2, 0].sum() > 0
    if weights[1]:
        assert o.jstat[..., 1, 1].sum() > 0                       # AM proposals were made and accepted
        piece = C.c_int32(0)
        _lib.check(g.lib.ptmi_split_am_piece(g.h, C.byref(piece)))
        assert piece.value >= 1                                   # ... on the row kernels, their increments from the matrix cores


def test_accept_propose_refuses_a_swap_iteration(mods):
    orc, _lib, PTEngine = mods
    import torch
    g = PTEngine(10, 3, 4, np.eye(10) * 0.01, weights=(20, 0, 0), tskip=5, split=True)
    g.init_state(np.zeros(10))
    z = torch.zeros((4, 3), dtype=torch.float64, device=g.device)
    _lib.check(g.lib.ptmi_propose(g.h, 5))
    with pytest.raises(_lib.PtmiError, match="swap iteration"):
        _lib.check(g.lib.ptmi_accept_propose(g.h, 5, z.data_ptr(), z.data_ptr()))


def test_full_size_callback_path_one_launch_equals_two_and_holds_its_invariants(mods):
    """BASELINE configs[1] at full size through the callback path as bench.py --callback times it: 64 x 4096 x 100-d, SCAM cycle,
    pooled covariance, one proposal launch + per iteration [torch callback, ptmi_accept_propose], through two covariance epochs and
    four swap epochs.  The callback's sum has torch's order, so the chains are not the fused kernel's bit for bit (the small cases
    above are, with the oracle's likelihood bits); at this size: (1) one launch per iteration equals two, every buffer of the whole
    batch bit for bit; (2) lnL is the callback's value of the row every chain holds; (3) the accept bookkeeping is consistent and
    the ring row of the last iteration is the cold chains' state; (4) a proposal moves a state along exactly one row of the
    eigenvector table (PTMCMCSampler.py:868-873)."""
    import torch
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    logl, _ = _callbacks(torch)
    kw = dict(weights=(20, 0, 0), cov_update=100, burn=10000, tskip=50, seed=1234, cov_mode="pooled", split=True)
    runs = []
    for fused in (True, False):
        g = PTEngine(d, nt, W, cov0, **kw)
        g.init_state_callback(np.zeros(d), logl, None)
        g.run_callback(230, logl, None, fused=fused)             # two covariance epochs, four swap epochs
        g.sync()
        runs.append(g)
    a, b = runs
    for k in ("X", "lnL", "lp", "slot_of", "nacc", "jstat", "nswap", "AM", "cov", "Ut"):
        assert torch.equal(a.t[k], b.t[k]), k
    assert torch.equal(a.proposals(), b.proposals()) and not a.t["sloc"].any()
    X, lnL = a.t["X"], a.t["lnL"]
    assert torch.equal(lnL, logl(X.view(-1, d)).view(W, nt))      # the value the callback returned for the held row
    js = a.get("jstat").astype(np.int64)
    assert (js[..., 0, 0] == 230).all() and (js[..., 0, 1] == a.get("nacc").astype(np.int64)).all()
    so = a.get("slot_of").astype(np.int64)
    cold = np.take_along_axis(a.get("X"), so[:, :1, None], 1)[:, 0]
    assert_same(a.get("AM")[:, 230 % 100], cold, "ring row of the last iteration = the cold chains' states")
    # a proposal moves the state along ONE eigen-direction (PT:868-873)
    _lib.check(a.lib.ptmi_propose(a.h, 231))
    a.sync()
    D = (a.proposals() - a.t["X"]).view(-1, d)[::97]
    Ut = a.t["Ut"][0, 0]
    coef = D @ Ut.T                                              # components along the table's rows
    k = coef.abs().argmax(1)
    rest = D - coef.gather(1, k[:, None]) * Ut[k]
    assert float(rest.abs().max()) <= 1e-12 * float(D.abs().max())
    assert a.eig_epochs == 2 and a.get("nswap").sum() > 0


@pytest.mark.parametrize("d,nt,W,weights,cov_mode", [(100, 4, 5, (20, 0, 20), "per_walker"), (130, 3, 4, (20, 0, 20), "pooled"),
                                                     (1000, 2, 3, (20, 0, 0), "pooled"), (20, 5, 3, (20, 0, 20), "per_walker")])
def test_callback_path_with_the_library_likelihood_equals_the_oracle(mods, d, nt, W, weights, cov_mode):
    """ptmi_rows_logl as the callback (the built-in isotropic Gaussian as a device kernel behind the C ABI, the fused kernels' lanes
    and summation order) at 4, 16 and 64 lanes per chain: the callback path IS the oracle's run, bit for bit."""
    orc, _lib, PTEngine = mods
    g, o = _pair(mods, d, nt, W, weights=weights, cov_update=25, burn=50, tskip=10, seed=3, split=True, cov_mode=cov_mode, am_mode="rows")
    logl = g.builtin_logl()
    g.init_state_callback(g.get("X"), logl, None)
    assert_same(g.get("lnL"), o.lnL, "initial lnL through ptmi_rows_logl")
    for n in (60, 7, 63):
        g.run_callback(n, logl, None)
        o.run(n)
        _compare(g, o, "library likelihood as callback, d=%d it=%d " % (d, g.iter))
        assert_same(g.get("cov"), o.cov, "cov")
        assert_same(g.get("Ut"), o.Ut, "Ut")
    assert o.nswap.sum() > 0


def test_full_size_callback_path_equals_the_fused_kernels_run(mods):
    """BASELINE configs[1] at full size, 64 x 4096 x 100-d, SCAM cycle, pooled covariance: the callback path (one proposal launch, then
    per iteration [ptmi_rows_logl on the proposals, ptmi_accept_propose]; two proposal buffers) against the FUSED step kernel's run of
    the same configuration -- every chain, counter, ring row, covariance and table of the whole batch bit for bit through two
    covariance epochs and four swap epochs (the fused run itself is held to the oracle by tests/test_gpu_bench_kernels.py)."""
    import torch
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    cov0 = np.eye(d) * 0.01
    kw = dict(weights=(20, 0, 0), cov_update=100, burn=10000, tskip=50, seed=1234, cov_mode="pooled", am_mode="rows")
    f = PTEngine(d, nt, W, cov0, **kw)
    f.init_state(np.zeros(d))
    f.run(230)
    g = PTEngine(d, nt, W, cov0, split=True, **kw)
    logl = g.builtin_logl()
    g.init_state_callback(np.zeros(d), logl, None)
    g.run_callback(230, logl, None)
    f.sync()
    g.sync()
    for k in ("X", "lnL", "lp", "temp_of", "slot_of", "nacc", "jstat", "nswap", "AM", "mu", "M2", "cov", "Ut", "S"):
        assert torch.equal(f.t[k], g.t[k]), k
    assert g.eig_epochs == 2 and g.get("nswap").sum() > 0 and not g.t["sloc"].any()
    acc = g.get("nacc").astype(np.float64).mean() / 230
    assert 0.5 < acc < 1.0


@pytest.mark.parametrize("d,nt,W", [(200, 4, 6), (1000, 3, 5)])
def test_dense_gaussian_as_a_gemm_callback(mods, d, nt, W):
    """PTEngine.dense_logl_callback: the dense Gaussian beyond the 104 parameters the built-in family keeps in LDS, as ONE matrix
    product per iteration for the whole batch on the split path (tests/test_simple.py:14-41 at larger ndim).  A callback's sums are
    its own: checked against NumPy's quadratic form, the chains against the accept rule (PTMCMCSampler.py:605-622) on those values."""
    import torch
    orc, _lib, PTEngine = mods
    rs = np.random.RandomState(d)
    A = rs.randn(d, d)
    C_ = A @ A.T / d + 0.5 * np.eye(d)
    P, mu = np.linalg.inv(C_), rs.randn(d) * 0.1
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 20, 20), cov_update=30, burn=60, tskip=10, seed=4, split=True, cov_mode="pooled")
    cb = g.dense_logl_callback(mu, P)
    g.init_state_callback(mu + rs.randn(W, nt, d) * 0.05, cb, None)
    g.run_callback(150, cb, None)
    g.sync()
    X, lnL = g.get("X"), g.get("lnL")
    R = X - mu
    want = -0.5 * np.einsum("wti,ij,wtj->wt", R, P, R)
    assert np.allclose(lnL, want, rtol=1e-10, atol=1e-10)
    assert torch.equal(g.t["lnL"].view(-1), cb(g.t["X"].view(-1, d)))         # the callback's own value of the row every chain holds
    js = g.get("jstat").astype(np.int64)
    assert (js[..., :3, 0].sum(-1) == 150).all() and js[..., 1, 1].sum() > 0 and js[..., 2, 0].sum() > 0
    assert 0 < g.get("nacc").sum() < 150 * nt * W and g.get("nswap").sum() > 0 and g.eig_epochs >= 4


@pytest.mark.parametrize("kind,d,nt,W,weights,extra", [("hip", 100, 8, 5, (20, 0, 20), dict(cov_mode="pooled")),
                                                      ("torch", 100, 4, 7, (20, 0, 0), {}),
                                                      ("torch", 37, 5, 3, (20, 0, 20), dict(box=True)),
                                                      ("hip", 130, 3, 4, (20, 0, 20), dict(pick_mode="walker"))])
def test_callback_segments_as_graph_launches_change_nothing(mods, kind, d, nt, W, weights, extra):
    """PTEngine.run_callback(graph=True): every segment's launches -- one proposal launch, then per iteration the callback and
    ptmi_accept_propose, ptmi_accept at the end -- captured ONCE per segment length in a hipGraph (the iteration from a counter in
    device memory: ptmi_device_iter) and replayed.  Covariance epochs, the DE epoch and activation (the captured launches bake the DE
    ring's head in: captured again), swaps and segments of several lengths in between; every buffer equals the run without graphs."""
    import torch
    orc, _lib, PTEngine = mods
    extra = dict(extra)
    box = extra.pop("box", False)
    rs = np.random.RandomState(d)
    lo, hi = (-0.4 - 0.1 * rs.rand(d), 0.4 + 0.1 * rs.rand(d)) if box else (None, None)
    p0 = rs.randn(W, nt, d) * 0.05
    kw = dict(weights=weights, cov_update=20, burn=40, tskip=7, seed=31, split=True, **extra)
    runs = []
    for graph in (True, False):
        g = PTEngine(d, nt, W, np.eye(d) * 0.01, **kw)
        logl, logp = _callbacks(torch, lo, hi)
        if kind == "hip":
            logl = g.builtin_logl()
        g.init_state_callback(p0, logl, logp)
        for n in (25, 3, 1, 46, 30):
            g.run_callback(n, logl, logp, graph=graph)
        runs.append(_snapshot(g))
        if graph:
            assert len(g._graphs) >= 4                               # several segment lengths, DE off / on, moving ring heads
    _same(runs[0], runs[1], "graph launches vs plain launches")
    assert runs[0]["nacc"].sum() > 0


def test_full_size_callback_path_samples_the_target_at_every_temperature(mods):
    """BASELINE configs[1] through the callback path with a TORCH callback (its own summation order, so no bit-for-bit partner), 30 000
    iterations from p0 = 0: 300 segments, 30 covariance epochs, two proposal buffers changing roles every iteration.  Rank t of an
    isotropic Gaussian at temperature T_t holds x ~ N(0, T_t I): <lnL> = -d T_t / 2 with standard deviation sqrt(d / 2) T_t across the
    4096 independent walkers for every rank up to T = 100 (above, the reference does not scale its jumps with sqrt(T),
    PTMCMCSampler.py:861-862), and the cold chains have mean 0 and variance 1 in every parameter -- the same property the fused path is
    held to (tests/test_gpu_bench_kernels.py)."""
    import torch
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 4096
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=17, cov_mode="pooled", split=True,
                 eig_lag=1)

    def logl(X):
        return torch.linalg.vector_norm(X, dim=-1).square_().mul_(-0.5)

    g.init_state_callback(np.zeros(d), logl, None)
    g.run_callback(30000, logl, None)
    g.sync()
    assert not g.t["sloc"].any()
    lnL, T = g.by_temp("lnL"), g.ladder
    warm = T <= 100.0
    z = (lnL.mean(0) + 0.5 * d * T) / (lnL.std(0) / np.sqrt(W))
    assert np.abs(z[warm]).max() < 5.0, z[warm]
    assert np.allclose(lnL.std(0)[warm], np.sqrt(d / 2.0) * T[warm], rtol=0.06)
    X = g.by_temp("X")[:, 0]
    assert np.abs(X.mean(0)).max() < 5.0 / np.sqrt(W)
    assert np.abs(X.var(0) - 1.0).max() < 5.0 * np.sqrt(2.0 / W)
    S = g.get("S")[0, 0]
    assert 0.9 < S.min() and S.max() < 1.1
    assert np.allclose(g.get("lnL"), -0.5 * (g.get("X") ** 2).sum(-1), rtol=1e-12, atol=1e-12)
    acc = g.get("nswap").astype(np.float64)[:, :nt - 1].mean(0) / g.swap_proposed
    assert 0.2 < acc[:30].min() and acc[:30].max() < 0.8


def test_full_size_callback_path_default_cycle_samples_the_target(mods):
    """The same property with the reference's DEFAULT cycle (SCAM / AM / DE 20 / 20 / 20) on the callback path: the AM picks' increments
    come from the matrix cores ahead of the proposals (ptmi_split_am_prepare, pieces of ten iterations), DE joins after burn = 2000 and
    reads its history rows in the row kernel.  64 x 2048 x 100-d, 16 000 iterations."""
    import torch
    orc, _lib, PTEngine = mods
    d, nt, W = 100, 64, 2048
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 20, 20), cov_update=1000, burn=2000, tskip=100, seed=23, cov_mode="pooled", split=True,
                 eig_lag=1)
    logl = g.builtin_logl()
    g.init_state_callback(np.zeros(d), logl, None)
    g.run_callback(16000, logl, None)
    g.sync()
    assert g.de_on and not g.t["sloc"].any()
    js = g.get("jstat").astype(np.int64)
    assert (js[..., :3, 0].sum(-1) == 16000).all() and (js[..., :3, 1] > 0).all()              # every type proposed and accepted on every rank
    lnL, T = g.by_temp("lnL"), g.ladder
    warm = T <= 100.0
    z = (lnL.mean(0) + 0.5 * d * T) / (lnL.std(0) / np.sqrt(W))
    assert np.abs(z[warm]).max() < 5.0, z[warm]
    assert np.allclose(lnL.std(0)[warm], np.sqrt(d / 2.0) * T[warm], rtol=0.08)
    X = g.by_temp("X")[:, 0]
    assert np.abs(X.mean(0)).max() < 5.0 / np.sqrt(W)
    assert np.abs(X.var(0) - 1.0).max() < 5.0 * np.sqrt(2.0 / W)
