"""Device-side NUTS / HMC (csrc/ptmi_gj.inc.h) against the oracle's restatement of nutsjump.py, bit for bit.
The oracle functions are pinned to the reference in tests/test_gradjump.py (replay of the reference's draws)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(d, nt, W, cov0, **kw):
    from oracle import oracle as orc
    from ptmcmcsampler_amd.engine import PTEngine
    g, o = PTEngine(d, nt, W, cov0, **kw), orc.OracleEngine(d, nt, W, cov0, **kw)
    return g, o


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        ok = np.array_equal(a.view(np.uint64), b.astype(np.float64).view(np.uint64))
    else:
        ok = np.array_equal(a.astype(np.int64), b.astype(np.int64))
    if not ok:
        bad = np.argwhere(~((a == b) | ((a != a) & (b != b))))
        raise AssertionError("%s differs at %d places, first %s: %r vs %r" % (what, len(bad), bad[:1], a[tuple(bad[0])], b[tuple(bad[0])]))


CASES = [
    dict(d=5, nt=3, W=6, logl=("iso",), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10)),
    dict(d=20, nt=4, W=5, logl=("curved",), logp=("box", -10.0, 10.0), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.08, 2, 50)),
    dict(d=8, nt=2, W=4, logl=("dense",), logp=("flat",), grad_weights=(20, 0), weights=(5, 0, 0)),
    dict(d=32, nt=2, W=3, logl=("iso",), logp=("box", -3.0, 3.0), grad_weights=(5, 20), weights=(10, 0, 10)),
    dict(d=2, nt=3, W=7, logl=("curved",), logp=("box", -10.0, 10.0), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.08, 2, 50)),
    # wider lane layouts: 16 lanes per chain (33 <= d <= 112), 64 lanes (d <= 512)
    dict(d=40, nt=2, W=3, logl=("dense",), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10)),
    dict(d=100, nt=2, W=2, logl=("iso",), logp=("box", -4.0, 4.0), grad_weights=(10, 5), weights=(10, 0, 10)),
    dict(d=150, nt=2, W=2, logl=("iso",), logp=("flat",), grad_weights=(10, 10), weights=(10, 0, 0)),
    dict(d=34, nt=2, W=3, logl=("curved",), logp=("box", -10.0, 10.0), grad_weights=(10, 10), weights=(10, 0, 10), hmc=(0.08, 2, 50)),
    # tree-height caps that ARE reached (the reference has none, nutsjump.py:716-802; the default 24 never binds): the capped
    # call must end exactly as the oracle's, in the whole-wave layout (d <= 32) and in the per-chain one
    dict(d=6, nt=2, W=5, logl=("iso",), logp=("flat",), grad_weights=(20, 0), weights=(5, 0, 0), nuts_maxdepth=2),
    dict(d=20, nt=3, W=4, logl=("curved",), logp=("box", -10.0, 10.0), grad_weights=(20, 0), weights=(5, 0, 5), nuts_maxdepth=3),
    dict(d=40, nt=2, W=3, logl=("iso",), logp=("flat",), grad_weights=(20, 0), weights=(5, 0, 0), nuts_maxdepth=1),
    dict(d=5, nt=2, W=3, logl=("iso",), logp=("flat",), grad_weights=(20, 0), weights=(5, 0, 0), nuts_maxdepth=0),
    # DIAGONAL initial covariances: the whitening products (nutsjump.py:53-54, 71-90) are d multiplications (oracle tab_vec; libptmi
    # decides at ptmi_create) -- whole-wave layout (d <= 32), 16 and 64 lanes per chain; HMC and NUTS
    dict(d=20, nt=4, W=5, logl=("curved",), logp=("box", -10.0, 10.0), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.08, 2, 50), diag=True),
    dict(d=7, nt=3, W=4, logl=("iso",), logp=("flat",), grad_weights=(20, 5), weights=(5, 0, 5), diag=True),
    dict(d=40, nt=2, W=3, logl=("dense",), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), diag=True),
    dict(d=150, nt=2, W=2, logl=("iso",), logp=("flat",), grad_weights=(10, 10), weights=(10, 0, 0), diag=True),
    # the interval family (include/ptmi.h PTMI_LOGL_INTERVAL): the reference's own NUTS workload -- tests/test_nuts.py:173-221: 40-d, box
    # (0, 10), SCAM = AM = DE = NUTS = HMC = 10, HMCsteps = 100, HMCstepsize = 0.4 -- and the other layouts: two chains per wave
    # (diagonal covariance), one per wave, 64 lanes per chain; without gradient jumps (the family still lives in their kernel shapes)
    dict(d=40, nt=2, W=3, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.4, 2, 100)),
    dict(d=8, nt=3, W=5, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.2, 2, 12), diag=True),
    dict(d=20, nt=2, W=4, logl=("interval", -2.0, 3.0), logp=("flat",), grad_weights=(20, 5), weights=(5, 0, 5)),
    dict(d=130, nt=2, W=2, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(10, 10), weights=(10, 0, 0), hmc=(0.2, 2, 20), diag=True),
    dict(d=40, nt=3, W=4, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(0, 0), weights=(10, 10, 10)),
    # the 16-lane shape at ndim <= 64 with a diagonal covariance: a gradient jump takes the whole wave, one element per lane, the chain's 16
    # lane groups in the wave's quads (GradJumpWide<16, L, 16>; the reference's test_nuts covariance -- from the Hessian of a target that
    # factorizes -- is diagonal); the boundaries 33 and 64, a box prior, the curved family's partner lanes, a tree cap
    dict(d=40, nt=2, W=5, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.4, 2, 100), diag=True),
    dict(d=33, nt=3, W=3, logl=("iso",), logp=("box", -3.0, 3.0), grad_weights=(20, 5), weights=(5, 0, 5), diag=True),
    dict(d=64, nt=2, W=3, logl=("iso",), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 0), diag=True),
    dict(d=34, nt=2, W=3, logl=("curved",), logp=("box", -10.0, 10.0), grad_weights=(10, 10), weights=(10, 0, 10), hmc=(0.08, 2, 50), diag=True),
    dict(d=50, nt=2, W=3, logl=("iso",), logp=("flat",), grad_weights=(20, 0), weights=(5, 0, 0), nuts_maxdepth=2, diag=True),
    dict(d=40, nt=2, W=3, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(10, 10), weights=(10, 0, 0), diag=True, nowide16=True),
    # AM increments ahead of the launch in the gradient-jump kernels (16 / 64 lanes per chain, pooled covariance: am_gemm_kernel, round 5),
    # a 1 MB scratch cutting the launches into single steps -- the per-chain layout and the whole-wave one
    dict(d=100, nt=2, W=3, logl=("iso",), logp=("box", -4.0, 4.0), grad_weights=(10, 5), weights=(10, 10, 10), cov_mode="pooled", am_mode="rows", am_budget=1),
    dict(d=40, nt=2, W=4, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.4, 2, 100), diag=True, cov_mode="pooled", am_mode="rows", am_budget=1),
    dict(d=150, nt=2, W=2, logl=("iso",), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 0), cov_mode="pooled", am_mode="rows"),
    # the launch order: chains with a wave of their own (PTMI_GJ_SOLO: here 5 of 20 / 9 of 12 chains, and all of them), empty chain slots beside
    # them -- in the pair layout, the whole-wave layout and the per-chain one
    dict(d=20, nt=4, W=5, logl=("curved",), logp=("box", -10.0, 10.0), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.08, 2, 50), diag=True, solo=5),
    dict(d=5, nt=3, W=4, logl=("iso",), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), solo=9),
    dict(d=40, nt=2, W=3, logl=("dense",), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), solo=6),
    dict(d=8, nt=3, W=5, logl=("interval", 0.0, 10.0), logp=("flat",), grad_weights=(10, 10), weights=(10, 10, 10), hmc=(0.2, 2, 12), diag=True, solo=0),
    dict(d=7, nt=3, W=4, logl=("interval", -1.0, 1.0), logp=("flat",), grad_weights=(0, 0), weights=(10, 0, 10)),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "d%d-%s%s%s%s" % (c["d"], c["logl"][0], "-cap%d" % c["nuts_maxdepth"] if "nuts_maxdepth" in c else "",
                                                                       "-diag" if c.get("diag") else "", ("" if sum(c["grad_weights"]) else "-nogj") + ("-nowide16" if c.get("nowide16") else "") + ("-solo%d" % c["solo"] if "solo" in c else "") + ("-" + c["cov_mode"] if "cov_mode" in c else "") + ("-pieces" if "am_budget" in c else "")))
def test_device_gradient_jumps_bit_exact(case, monkeypatch):
    c = dict(case)
    d, nt, W = c.pop("d"), c.pop("nt"), c.pop("W")
    diag = c.pop("diag", False)
    if "am_budget" in c:
        monkeypatch.setenv("PTMI_AM_BUDGET_MB", str(c.pop("am_budget")))
    if "solo" in c:
        monkeypatch.setenv("PTMI_GJ_SOLO", str(c.pop("solo")))
    if c.pop("nowide16", False):
        monkeypatch.setenv("PTMI_GJ_NOWIDE16", "1")               # the per-chain layout with the same diagonal tables
    if diag and d == 7:
        monkeypatch.setenv("PTMI_GJ_NOPAIR", "1")                 # the one-chain-per-wave layout with diagonal tables (the default pairs two chains per wave)
    rs = np.random.RandomState(d)
    if c["logl"][0] == "dense":
        A = rs.randn(d, d)
        c["logl"] = ("dense", rs.randn(d) * 0.1, np.linalg.inv(A @ A.T / d + 0.5 * np.eye(d)))
    if c["logp"][0] == "box":
        c["logp"] = ("box", np.full(d, c["logp"][1]), np.full(d, c["logp"][2]))
    if c["logl"][0] == "interval":                             # uneven boxes around the case's (a, b)
        c["logl"] = ("interval", c["logl"][1] - rs.uniform(0, 0.5, d), c["logl"][2] + rs.uniform(0, 2.0, d))
    A = rs.randn(d, d)
    cov0 = (A @ A.T / d + np.eye(d)) * (1.0 if c["logl"][0] == "curved" else 0.3)
    if diag:
        cov0 = np.diag(np.diag(cov0))
    kw = dict(cov_update=50, burn=100, tskip=10, seed=31, **c)
    g, o = _pair(d, nt, W, cov0, **kw)
    p0 = rs.randn(W, nt, d) * 0.3
    if c["logl"][0] == "curved":
        p0 = np.tile(np.array([-0.1, -0.5] * (d // 2)), (W, nt, 1)) + rs.randn(W, nt, d) * 0.05
    g.init_state(p0)
    o.init_state(p0)
    n = 260
    g.run(n)
    o.run(n)
    g.sync()
    for name in ("X", "lnL", "lp", "temp_of", "slot_of", "nacc", "jstat", "nswap", "AM"):
        _same(g.get(name), getattr(o, name), name)
    if sum(c["grad_weights"]):
        _same(g.get("gj"), o.gj, "gradient-jump state")
    js = o.jstat.sum(axis=(0, 1))
    if c["grad_weights"][0]:
        assert js[3, 0] > 0 and js[3, 0] == js[3, 1], "NUTS proposals are always accepted (NJ:838)"
        assert (o.gj[..., 4] > 0).all()
    if c["grad_weights"][1]:
        assert js[4, 0] > 0
    assert js[:, 0].sum() == W * nt * n
    if "nuts_maxdepth" in c:                                   # the cap did bind: without it the same run takes more leapfrogs
        from oracle import oracle as orc
        free = orc.OracleEngine(d, nt, W, cov0, **dict(kw, nuts_maxdepth=24))
        free.init_state(p0)
        free.run(n)
        assert free.gj[..., 7].sum() > o.gj[..., 7].sum() > 0


def test_sharded_ladder_with_gradient_jumps():
    """Config-5 shape in small: curved likelihood, box prior, DE + SCAM + NUTS + HMC, ladder sharded over two
    emulated ranks on one GPU == the oracle's single-process run."""
    import os
    import sys
    import threading
    sys.path.insert(0, os.path.dirname(__file__))
    from thread_comm import ThreadComm, ThreadWorld
    from oracle import oracle as orc
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    d, ntb, W, n, nranks = 20, 4, 6, 230, 2
    ntg = ntb * nranks
    kw = dict(logl=("curved",), logp=("box", np.full(d, -10.0), np.full(d, 10.0)), weights=(10, 0, 10), grad_weights=(10, 10),
              hmc=(0.08, 2, 50), cov_update=50, burn=100, tskip=10, seed=9, cov_mode="per_walker")
    cov0 = np.eye(d)
    p0 = np.tile(np.array([-0.1, -0.5] * (d // 2)), (W, ntg, 1)) + np.random.RandomState(1).randn(W, ntg, d) * 0.05
    ref = orc.OracleEngine(d, ntg, W, cov0, **kw)
    ref.init_state(p0)
    ref.run(n)
    world = ThreadWorld(nranks)
    engines, errs = [None] * nranks, []

    def rank_main(r):
        try:
            e = ShardedPTEngine(d, ntg, W, cov0, comm=ThreadComm(world, r), **kw)
            engines[r] = e
            e.init_state(p0)
            e.run(n)
            e.sync()
        except BaseException as ex:  # noqa
            errs.append(ex)
            world.bar.abort()
            raise

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for r, e in enumerate(engines):
        L, sl = e.local, slice(r * ntb, (r + 1) * ntb)
        so = L.get("slot_of")
        bt = lambda a: np.take_along_axis(a, so.reshape(so.shape + (1,) * (a.ndim - 2)), axis=1)  # noqa: E731
        _same(bt(L.get("X")), ref.by_temp(ref.X)[:, sl], "X")
        _same(L.get("gj"), ref.gj[:, sl], "gradient-jump state")
        _same(L.get("jstat"), ref.jstat[:, sl], "jstat")
    assert ref.jstat[..., 3, 0].sum() > 0 and ref.nswap.sum() > 0


def test_facade_runs_device_gradient_jumps(tmp_path):
    """PTSampler with a device likelihood and logl_grad=True: NUTS / HMC enter the cycle under the reference's jump
    names, their statistics and files are written, NUTS is always accepted."""
    from ptmcmcsampler_amd import PTSampler
    d = 4
    s = PTSampler(d, ("curved",), ("box", np.full(d, -10.0), np.full(d, 10.0)), np.eye(d), outDir=str(tmp_path), ntemps=3, nwalkers=4,
                  logl_grad=True, logp_grad=True, seed=5, verbose=False)
    s.sample(np.array([-0.1, -0.5] * (d // 2)), 400, burn=100, covUpdate=100, thin=1, isave=100, Tskip=10, SCAMweight=10, AMweight=10,
             DEweight=10, NUTSweight=10, HMCweight=10, MALAweight=0, HMCsteps=50, HMCstepsize=0.08)
    assert s.jumpDict["NUTSJUMP"][0] > 0 and s.jumpDict["NUTSJUMP"][0] == s.jumpDict["NUTSJUMP"][1]
    assert s.jumpDict["HMCJump"][0] > 0
    assert sum(v[0] for v in s.jumpDict.values()) == 400
    for name in ("NUTSJUMP_jump.txt", "HMCJump_jump.txt", "covarianceJumpProposalSCAM_jump.txt", "chain_1.0.txt"):
        assert (tmp_path / name).exists(), name
    chain = np.loadtxt(tmp_path / "chain_1.0.txt")
    assert chain.shape[1] == d + 4 and len(chain) in (400, 401) and np.isfinite(chain).all()


def test_tree_levels_beyond_the_lds_budget_live_in_global_scratch():
    """The tree stack keeps the low heights in LDS and the rest -- heights 11 .. 24 by default, never reached by a sane run --
    in global scratch.  With PTMI_GJ_LDS_LEVELS=1 (read once per process, hence the child) every height above 0 takes the
    global path: the same parity cases must still hold bit for bit, in the whole-wave layout and in the per-chain one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gradjump_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "test_device_gradient_jumps_bit_exact and (d20-curved or d5-iso or d40-dense or d2-curved)"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PTMI_GJ_LDS_LEVELS="1"), cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
