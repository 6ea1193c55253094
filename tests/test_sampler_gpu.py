"""PTSampler facade on the GPU: the reference's script surface (tests/test_simple.py of the reference is
the workload model), output files and statistics.  Trajectories are not comparable with the reference
here (different RNG); the oracle tests pin those."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class GaussianLikelihood(object):
    """The workload of the reference's tests/test_simple.py:14-41 (seeded here)."""

    def __init__(self, ndim=2, pmin=-10, pmax=10, seed=0):
        rs = np.random.RandomState(seed)
        self.a = np.ones(ndim) * pmin
        self.b = np.ones(ndim) * pmax
        self.mu = rs.uniform(pmin, pmax, ndim)
        cov = 0.5 - rs.rand(ndim ** 2).reshape((ndim, ndim))
        cov = np.triu(cov)
        cov += cov.T - np.diag(cov.diagonal())
        self.cov = np.dot(cov, cov)
        self.icov = np.linalg.inv(self.cov)

    def lnlikefn(self, x):
        diff = x - self.mu
        return -np.dot(diff, np.dot(self.icov, diff)) / 2.0

    def lnpriorfn(self, x):
        if np.all(self.a <= x) and np.all(self.b >= x):
            return 0.0
        return -np.inf


class UniformJump(object):
    def __init__(self, pmin, pmax, seed=1):
        self.pmin, self.pmax, self.rs = pmin, pmax, np.random.RandomState(seed)

    def jump(self, x, it, beta):
        return self.rs.uniform(self.pmin, self.pmax, len(x)), 0


def test_reference_script_surface_with_python_callbacks(tmp_path):
    """tests/test_simple.py of the reference, through the facade: Python logl/logp + a custom jump."""
    from ptmcmcsampler_amd import PTSampler
    ndim, pmin, pmax = 4, 0.0, 10.0
    glo = GaussianLikelihood(ndim, pmin, pmax, seed=3)
    p0 = np.random.RandomState(4).uniform(pmin, pmax, ndim)
    cov = np.eye(ndim) * 0.1 ** 2
    s = PTSampler(ndim, glo.lnlikefn, glo.lnpriorfn, np.copy(cov), outDir=str(tmp_path), verbose=False, seed=5)
    s.addProposalToCycle(UniformJump(pmin, pmax).jump, 5)
    s.sample(p0, 6000, burn=500, thin=1, covUpdate=500, SCAMweight=20, AMweight=20, DEweight=20)
    assert s._chain.shape == (6001, ndim)
    assert np.array_equal(s._chain[0], p0) and np.isclose(s._lnlike[0], glo.lnlikefn(p0))
    assert set(s.jumpDict) == {"jump", "covarianceJumpProposalSCAM", "covarianceJumpProposalAM", "DEJump"}
    assert sum(v[0] for v in s.jumpDict.values()) == 6000 and s.jumpDict["jump"][0] > 300
    assert 0.05 < s.naccepted / 6000 < 0.9
    # lnlike column is the callback's value of the stored sample
    for i in (1, 777, 6000):
        assert np.isclose(s._lnlike[i], glo.lnlikefn(s._chain[i]), rtol=1e-12)
    # posterior of the T=1 chain ~ N(mu, cov) (the box is wide): loose statistical check after burn-in
    x = s._chain[1500:]
    sd = np.sqrt(np.diag(glo.cov))
    assert np.all(np.abs(x.mean(0) - glo.mu) < 0.5 * sd + 0.05)
    # files: chain_1.txt (single chain: integer ladder), cov.npy, jumps.txt, <name>_jump.txt
    names = set(os.listdir(tmp_path))
    assert {"chain_1.txt", "cov.npy", "jumps.txt", "jump_jump.txt", "DEJump_jump.txt"} <= names
    rows = open(tmp_path / "chain_1.txt").read().splitlines()
    assert len(rows) == 6001
    cols = rows[0].split("\t")
    assert len(cols) == ndim + 4 and all(re.fullmatch(r"-?\d+\.\d{22}", c) for c in cols[:ndim])
    assert np.allclose([float(c) for c in cols[:ndim]], p0)
    fr = dict(l.split() for l in open(tmp_path / "jumps.txt").read().splitlines())
    assert abs(float(fr["jump"]) - 5 / 65) < 0.01 and abs(float(fr["DEJump"]) - 20 / 65) < 0.01
    assert np.load(tmp_path / "cov.npy").shape == (ndim, ndim)
    assert s.cov is not cov and not np.array_equal(s.cov, cov)        # adapted in place


def test_first_chain_row_matches_reference_file_format(tmp_path, golden):
    """Row 0 of chain_<T>.txt is p0, lnprob, lnlike, 0, 1 in the reference's formats (:741-745)."""
    from ptmcmcsampler_amd import PTSampler
    g = golden("traj_pt4_d6")
    d = int(g["ndim"])
    s = PTSampler(d, ("iso",), ("flat",), np.copy(g["cov0"]), outDir=str(tmp_path), verbose=False, seed=1, ntemps=4)
    s.sample(g["p0"], 100, covUpdate=50, burn=100, thin=1, isave=50, Tskip=10)
    assert str(g["chainfile_name"]) == "chain_1.0.txt" and os.path.exists(tmp_path / "chain_1.0.txt")
    mine = open(tmp_path / "chain_1.0.txt").read().splitlines()
    ref0 = str(g["chainfile_head"][0]).split("\t")
    got0 = mine[0].split("\t")
    assert got0[:d] == ref0[:d]                       # "%22.22f" of p0, character for character
    assert got0[d:d + 2] == ref0[d:d + 2]             # lnprob, lnlike of p0 ("%f")
    assert got0[d + 2:] == ref0[d + 2:] == ["0.000000", "1.000000"]
    assert len(mine) == 101


def test_fused_path_statistics_dense_gaussian(tmp_path):
    """Device likelihood + fused kernel: 16 walkers x 4 temperatures on a 10-d dense Gaussian."""
    from ptmcmcsampler_amd import PTSampler
    d = 10
    rs = np.random.RandomState(2)
    A = rs.randn(d, d)
    C = A @ A.T / d + 0.3 * np.eye(d)
    mu = rs.randn(d)
    s = PTSampler(d, ("dense", mu, np.linalg.inv(C)), ("flat",), np.eye(d) * 0.01, outDir=str(tmp_path), verbose=False,
                  seed=11, ntemps=4, nwalkers=16, keep_walkers=16)
    s.sample(mu + 0.1, 20000, burn=2000, thin=10, covUpdate=1000, isave=1000, Tskip=100)
    x = s._chains[:, 300:, :].reshape(-1, d)
    assert x.shape[0] == 16 * 1701
    se = np.sqrt(np.diag(C) / (x.shape[0] / 40.0))
    assert np.all(np.abs(x.mean(0) - mu) < 5 * se)
    emp = np.cov(x.T)
    assert np.max(np.abs(emp - C)) / np.max(np.abs(C)) < 0.15
    assert s.swapProposed == 200 and s.nswap_accepted > 0
    assert os.path.exists(tmp_path / "chain_1.0.txt") and os.path.exists(tmp_path / "chain_1.0_w15.txt")
    assert len(open(tmp_path / "chain_1.0.txt").read().splitlines()) == 2001


def test_argument_errors(tmp_path):
    from ptmcmcsampler_amd import PTSampler
    s = PTSampler(3, ("iso",), ("flat",), np.eye(3), outDir=str(tmp_path), verbose=False)
    with pytest.raises(ValueError, match="isave = 1000 is not a multiple of thin =  7"):
        s.sample(np.zeros(3), 100, thin=7)
    s2 = PTSampler(3, ("iso",), ("flat",), np.eye(3), outDir=str(tmp_path), verbose=False)
    with pytest.raises(ValueError, match="No jump proposals specified!"):
        s2.sample(np.zeros(3), 100, SCAMweight=0, AMweight=0, DEweight=0)


def test_gradient_jumps_through_the_facade(tmp_path, capsys):
    """logl_grad / logp_grad given: HMC and NUTS join the cycle as in the reference (PTMCMCSampler.py:226-258)."""
    from ptmcmcsampler_amd import PTSampler
    d = 4
    rs = np.random.RandomState(8)
    A = rs.randn(d, d)
    C = A @ A.T / d + 0.4 * np.eye(d)
    P = np.linalg.inv(C)
    ll = lambda x: -0.5 * x @ P @ x                       # noqa: E731
    lp = lambda x: 0.0                                    # noqa: E731
    s = PTSampler(d, ll, lp, np.copy(C) * 0.5, logl_grad=lambda x: (ll(x), -P @ x), logp_grad=lambda x: (0.0, np.zeros(d)),
                  outDir=str(tmp_path), verbose=False, seed=3, ntemps=2)
    np.random.seed(10)
    s.sample(np.zeros(d), 1500, burn=300, thin=1, covUpdate=300, SCAMweight=10, AMweight=10, DEweight=10, NUTSweight=10,
             HMCweight=10, MALAweight=0, HMCstepsize=0.3, HMCsteps=20, Tskip=50)
    out = capsys.readouterr().out
    assert out.count("WARNING: GradientJumps not yet adaptive") == 2        # once per jump type
    assert {"HMCJump", "NUTSJUMP", "covarianceJumpProposalSCAM", "DEJump"} <= set(s.jumpDict)
    prop, acc = s.jumpDict["NUTSJUMP"]
    assert prop > 150 and acc / prop > 0.97               # NUTS proposals are constructed to be accepted
    x = s._chain[400:]
    assert np.max(np.abs(np.cov(x.T) - C)) / np.max(C) < 0.35


def test_the_references_nuts_test_on_the_device_family(tmp_path, capsys):
    """The reference's tests/test_nuts.py:173-221 with its likelihood as the device family ("interval", a, b) (include/ptmi.h
    PTMI_LOGL_INTERVAL): 40-d, box (0, 10), the covariance from the Hessian at the maximum (closed form here: the target factorizes),
    sample(p0, 1000, burn=500, thin=1, covUpdate=500, SCAM = AM = DE = NUTS = HMC = 10, HMCsteps=100, HMCstepsize=0.4) -- the
    reference's test only asks that this runs; here also: the chain, mapped back to x, is the unit Gaussian cut at 0 (mean
    sqrt(2/pi) = 0.798, second moment 1)."""
    import scipy.optimize as so
    from ptmcmcsampler_amd import PTSampler
    ndim, a, b = 40, 0.0, 10.0

    def lnl1(p):                                              # one coordinate of the target (test_gradjump._interval_callbacks has all of it)
        x = (b - a) * np.exp(p) / (1 + np.exp(p)) + a
        return -0.5 * x * x + p - 2 * np.log(1.0 + np.exp(p))

    pmax = so.minimize_scalar(lambda p: -lnl1(p), bounds=(-10, 5), method="bounded", options=dict(xatol=1e-12)).x
    h = (lnl1(pmax + 1e-4) - 2 * lnl1(pmax) + lnl1(pmax - 1e-4)) / 1e-8
    p0, cov = np.full(ndim, pmax), np.eye(ndim) / -h
    kw = dict(burn=500, thin=1, covUpdate=500, SCAMweight=10, AMweight=10, DEweight=10, NUTSweight=10, HMCweight=10, MALAweight=0,
              HMCsteps=100, HMCstepsize=0.4)
    s = PTSampler(ndim, ("interval", np.full(ndim, a), np.full(ndim, b)), ("flat",), np.copy(cov), logl_grad=True, logp_grad=True,
                  outDir=str(tmp_path / "dev"), verbose=False, seed=5)
    s.sample(np.copy(p0), 4000, **kw)
    assert {"HMCJump", "NUTSJUMP", "covarianceJumpProposalSCAM", "covarianceJumpProposalAM", "DEJump"} <= set(s.jumpDict)
    prop, acc = s.jumpDict["NUTSJUMP"]
    assert prop > 400 and acc / prop > 0.97
    chain = np.loadtxt(str(tmp_path / "dev" / "chain_1.txt"))
    assert chain.shape == (4001, ndim + 4)                    # row 0 is the start (PTMCMCSampler.py:479-493)
    x = (b - a) * np.exp(chain[800:, :ndim]) / (1 + np.exp(chain[800:, :ndim])) + a
    assert abs(x.mean() - np.sqrt(2 / np.pi)) < 0.03 and abs((x * x).mean() - 1.0) < 0.06
    assert np.all(np.abs(x.mean(0) - np.sqrt(2 / np.pi)) < 0.25)


def test_resume_continues_bit_identically(tmp_path):
    """resume=True: a run stopped after 600 iterations and resumed to 1200 equals one uninterrupted run,
    state and chain file alike (device checkpoint + counter-based RNG; the reference replays its text file)."""
    from ptmcmcsampler_amd import PTSampler
    d = 6
    kw = dict(burn=300, thin=2, covUpdate=100, isave=200, Tskip=20, SCAMweight=20, AMweight=20, DEweight=20)
    rs = np.random.RandomState(1)
    p0 = rs.randn(d) * 0.2

    def make(out, resume=False):
        return PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(out), verbose=False, seed=77, ntemps=3,
                         nwalkers=4, keep_walkers=2, resume=resume, checkpoint=True)
    a = make(tmp_path / "a")
    a.sample(p0, 1200, **kw)
    b1 = make(tmp_path / "b")
    b1.sample(p0, 600, **kw)
    b2 = make(tmp_path / "b", resume=True)
    b2.sample(p0, 1200, **kw)
    for name in ("X", "lnL", "temp_of", "nacc", "jstat", "nswap", "Ut", "DE", "M2"):
        assert np.array_equal(a.engine.get(name), b2.engine.get(name)), name
    assert np.array_equal(a._chains, b2._chains) and np.array_equal(a._lnprobs, b2._lnprobs)
    fa = open(tmp_path / "a" / "chain_1.0.txt").read().splitlines()
    fb = open(tmp_path / "b" / "chain_1.0.txt").read().splitlines()
    assert len(fa) == len(fb) == 601
    # the rate columns are "as of the time of writing" (PTMCMCSampler.py:741-745): identical in both runs
    assert fa == fb
    assert a.jumpDict == b2.jumpDict and a.naccepted == b2.naccepted


def test_write_hot_chains_and_groups(tmp_path):
    """writeHotChains / hotChain file naming (PTMCMCSampler.py:281-288) and parameter groups through the facade."""
    from ptmcmcsampler_amd import PTSampler
    d = 6
    groups = [np.array([0, 1, 2]), np.array([3, 4, 5])]
    s = PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, groups=groups, outDir=str(tmp_path), verbose=False, seed=9, ntemps=3)
    s.sample(np.zeros(d), 400, burn=100, thin=5, covUpdate=100, isave=100, Tskip=10, writeHotChains=True, hotChain=True)
    names = sorted(os.listdir(tmp_path))
    t1 = float(s.ladder[1])
    assert "chain_1.0.txt" in names and "chain_hot.txt" in names and "chain_{0}.txt".format(t1) in names
    for f in ("chain_1.0.txt", "chain_hot.txt", "chain_{0}.txt".format(t1)):
        rows = np.loadtxt(tmp_path / f)
        assert rows.shape == (81, d + 4)
        assert np.allclose(rows[:, d + 1], -0.5 * (rows[:, :d] ** 2).sum(1), atol=2e-6)       # lnlike column, "%f"
    hot = np.loadtxt(tmp_path / "chain_hot.txt")
    assert np.all(hot[:, d] == 0.0) or np.allclose(hot[1:, d], hot[1:, d + 1] / 1e80, atol=1e-6)   # lnprob = lnlike / 1e80
    assert len(s.U) == 2 and s.U[0].shape == (3, 3) and s.S[1].shape == (3,)
    assert np.abs(s.cov[0, 3]) < np.abs(s.cov[0, 0])          # adapted, and the facade mirrors the device covariance


def test_resume_without_checkpoint_refuses_to_truncate(tmp_path):
    """resume=True on a ladder whose device checkpoint is missing: the chain files must survive and the call must fail
    loudly (a text file can only be replayed for a single chain)."""
    from ptmcmcsampler_amd import PTSampler
    d = 3
    kw = dict(burn=100, thin=1, covUpdate=50, isave=100, Tskip=10)
    a = PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path), verbose=False, seed=1, ntemps=2)
    a.sample(np.zeros(d), 200, **kw)
    assert not os.path.exists(tmp_path / "ptmi_checkpoint.npz")                 # checkpoints are opt-in
    before = open(tmp_path / "chain_1.0.txt").read()
    b = PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path), verbose=False, seed=1, ntemps=2, resume=True)
    with pytest.raises(Exception, match="Couldn't resume"):
        b.sample(np.zeros(d), 400, **kw)
    assert open(tmp_path / "chain_1.0.txt").read() == before


def test_a_fresh_start_removes_an_older_runs_checkpoint(tmp_path):
    """Run 1 (checkpoint=True) leaves ptmi_checkpoint.npz.  Run 2 in the same outDir without resume truncates the chain file
    and must also remove that checkpoint: a later resume=True then replays run 2's chain file (one chain) instead of
    restoring run 1's device state over run 2's rows.  And a checkpoint of a different run is refused, not loaded."""
    from ptmcmcsampler_amd import PTSampler
    d = 3
    kw = dict(burn=100, thin=1, covUpdate=50, isave=100)

    def make(seed, **extra):
        return PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path), verbose=False, seed=seed, **extra)
    make(1, checkpoint=True).sample(np.zeros(d), 300, **kw)
    assert os.path.exists(tmp_path / "ptmi_checkpoint.npz")
    old_ckpt = open(tmp_path / "ptmi_checkpoint.npz", "rb").read()
    b = make(2)                                                                  # fresh start, no checkpoints
    b.sample(np.zeros(d), 200, **kw)
    assert not os.path.exists(tmp_path / "ptmi_checkpoint.npz")
    run2 = open(tmp_path / "chain_1.txt").read().splitlines()
    assert len(run2) == 201
    c = make(2, resume=True)                                                     # replays run 2's file (201 rows), continues to 400
    c.sample(np.zeros(d), 400, **kw)
    rows = open(tmp_path / "chain_1.txt").read().splitlines()
    assert len(rows) == 401 and rows[:201] == run2
    # a checkpoint that belongs to another run (seed 1) beside this run's chain file: refused by its fingerprint
    open(tmp_path / "ptmi_checkpoint.npz", "wb").write(old_ckpt)
    e = make(2, resume=True)
    with pytest.raises(Exception, match="different run .*seed"):
        e.sample(np.zeros(d), 600, **kw)
    assert open(tmp_path / "chain_1.txt").read().splitlines() == rows          # nothing was cut


def test_resume_cuts_every_chain_file_back_to_the_checkpoint(tmp_path):
    """A kill between the file write and the checkpoint leaves rows the resumed run writes again: cold, walker and hot
    files are all cut back to the checkpoint's row count (the hot files are written in the same call)."""
    from ptmcmcsampler_amd import PTSampler
    d = 4
    kw = dict(burn=100, thin=2, covUpdate=50, isave=100, Tskip=10, writeHotChains=True)

    def make(resume=False):
        return PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path), verbose=False, seed=5, ntemps=3, nwalkers=2,
                         keep_walkers=2, resume=resume, checkpoint=True)
    a = make()
    a.sample(np.zeros(d), 400, **kw)
    files = [f for f in sorted(os.listdir(tmp_path)) if f.startswith("chain_")]
    assert len(files) == 4                                                       # cold, cold of walker 1, two hot ranks
    good = {f: open(tmp_path / f).read() for f in files}
    for f in files:                                                              # the torn write: 7 more rows everywhere
        open(tmp_path / f, "a").write("".join(good[f].splitlines(True)[-7:]))
    b = make(resume=True)
    b.sample(np.zeros(d), 600, **kw)
    for f in files:
        rows = open(tmp_path / f).read().splitlines(True)
        assert len(rows) == 301 and "".join(rows[:201]) == good[f], f


def test_resume_from_a_chain_file_the_reference_wrote(tmp_path, golden, capsys):
    """PTMCMCSampler.py:290-319, 591-599: resume=True with nothing but the reference's chain_1.txt.  The rows are
    replayed as the chain's states; the covariance epochs and the DE history rebuilt from them equal the reference's own
    (its mu / M2 / cov snapshots and its DE buffer during its resumed run), bit for bit; sampling then continues and
    appends to the file."""
    from ptmcmcsampler_amd import PTSampler
    g = golden("resume")
    d = int(g["ndim"])
    open(tmp_path / str(g["chainfile_name"]), "w").write("\n".join(str(r) for r in g["chainfile"]) + "\n")
    kw = {k[3:]: int(g[k]) for k in g.files if k.startswith("kw_")}
    s = PTSampler(d, ("dense", g["mu"], g["icov"]), ("box", np.zeros(d), 10 * np.ones(d)), np.copy(g["cov0"]), outDir=str(tmp_path),
                  verbose=False, seed=8, resume=True)
    snaps = []
    replay = s._replay_chain_file

    def watched():
        eng = s.engine
        upd = eng.update_cov

        def update_cov(it_done):
            upd(it_done)
            snaps.append((it_done, eng.get("mu")[0].copy(), eng.get("M2")[0].copy(), eng.get("cov")[0].copy()))

        eng.update_cov = update_cov
        last = replay()
        eng.update_cov = upd
        snaps.append(("end", last, np.roll(eng.get("DE")[0], -eng.de_head, axis=0), eng.get("AM")[0].copy(), eng.get("X")[0, 0].copy(),
                      eng.get("lnL")[0, 0], int(eng.get("nacc")[0, 0])))
        return last

    s._replay_chain_file = watched
    s.sample(g["p0"], 1000, **kw)
    assert "Resuming with 301 samples from file representing 601 original samples" in capsys.readouterr().out
    end = snaps.pop()
    assert int(g["resume_length"]) == 301 and end[1] == 601
    nrep = sum(1 for i in range(int(g["nepochs"])) if int(g["ep_it_%d" % i]) <= 600)
    assert nrep == 6 and [sn[0] for sn in snaps] == [100, 200, 300, 400, 500, 600]
    for i, (it, mu, M2, cov) in enumerate(snaps):
        assert it == int(g["ep_it_%d" % i])
        for name, mine in (("mu", mu), ("M2", M2), ("cov", cov)):
            assert np.array_equal(mine.view(np.uint64), g["ep_%s_%d" % (name, i)].view(np.uint64)), (name, it)
    assert np.array_equal(end[2], g["replay_de"]) and np.array_equal(end[3], g["replay_am"])
    assert np.array_equal(end[4], g["replay_x"]) and end[5] == float(g["replay_lnl"])
    assert abs(end[6] - float(g["replay_nacc"])) <= 1.0
    rows = open(tmp_path / str(g["chainfile_name"])).read().splitlines()
    assert len(rows) == int(g["final_rows"]) == 501
    assert rows[:301] == [str(r) for r in g["chainfile"]]                          # the old rows are kept as they were
    new = np.loadtxt(tmp_path / str(g["chainfile_name"]))[301:]
    assert np.all(new[:, :d] >= 0) and np.all(new[:, :d] <= 10) and len(np.unique(new[:, 0])) > 20
    assert s.DEJump in s.propCycle and s.jumpDict["DEJump"][0] > 0


def test_resume_a_ladder_from_its_chain_files(tmp_path, capsys):
    """PTMCMCSampler.py:290-319, 591-599 for a LADDER: in the reference every MPI rank replays its own chain_<T>.txt, and the swaps
    of the replayed iterations still run on the replayed states (:624-627).  Here the one process replays all the files.  The
    procedure is checked against the oracle's pieces: AM rows = the cold file's rows, at swap iterations the state the
    oracle's sweep (same Philox uniforms) puts at rank 0; covariance epochs = the oracle's Welford over those rows; swap
    credits = the sweep's; then sampling continues and every file grows."""
    from oracle import oracle as orc
    from ptmcmcsampler_amd import PTSampler
    d, nt = 4, 3
    kw = dict(burn=200, thin=2, covUpdate=50, isave=100, Tskip=10, writeHotChains=True)
    p0 = np.full(d, 0.1)

    def make(resume):
        return PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path), verbose=False, seed=21, ntemps=nt, resume=resume,
                         checkpoint=False)
    a = make(False)
    a.sample(p0, 400, **kw)
    assert not os.path.exists(tmp_path / "ptmi_checkpoint.npz")
    names = ["chain_1.0.txt"] + ["chain_{0}.txt".format(a.ladder[r]) for r in range(1, nt)]
    old = {f: open(tmp_path / f).read() for f in names}
    rows = [np.loadtxt(tmp_path / f) for f in names]
    assert all(r.shape == (201, d + 4) for r in rows)
    b = make(True)
    snaps, end = [], {}
    replay = b._replay_chain_file

    def watched():
        eng = b.engine
        upd = eng.update_cov

        def update_cov(it_done):
            upd(it_done)
            snaps.append((it_done, eng.get("mu")[0].copy(), eng.get("M2")[0].copy(), eng.get("cov")[0].copy()))

        eng.update_cov = update_cov
        last = replay()
        eng.update_cov = upd
        eng.sync()
        end.update(last=last, AM=eng.get("AM")[0].copy(), X=eng.by_temp("X")[0].copy(), lnL=eng.by_temp("lnL")[0].copy(),
                   nswap=eng.get("nswap")[0].astype(np.int64).copy(), nacc=eng.get("nacc")[0].astype(np.int64).copy(), swaps=eng.swap_proposed)
        return last

    b._replay_chain_file = watched
    b.sample(p0, 600, **kw)
    assert "Resuming with 201 samples from file representing 401 original samples" in capsys.readouterr().out
    # the same procedure from the oracle's pieces
    thin, cu, last = kw["thin"], kw["covUpdate"], 201 * kw["thin"] - 1
    mu, M2, AM = np.zeros(d), np.zeros((d, d)), np.zeros((cu, d))
    nswap, nsw, want = np.zeros(nt, dtype=np.int64), 0, []
    AM[0] = rows[0][0, :d]
    for it in range(1, last + 1):
        if (it - 1) % cu == 0 and it - 1 != 0:
            cov = orc.welford(AM, mu, M2, it - 1)
            want.append((it - 1, mu.copy(), M2.copy(), cov.copy()))
        k = it // thin
        cold = rows[0][k, :d]
        if it % kw["Tskip"] == 0:
            lnl_pos = np.array([[rows[r][k, -3] for r in range(nt)]])
            m, acc = orc.swap_sweep(b.ladder, lnl_pos, it=it, seed=b.seed, walker0=0)
            cold = rows[int(m[0, 0])][k, :d]
            nswap += acc[0].astype(np.int64)
            nsw += 1
        AM[it % cu] = cold
    assert end["last"] == last == 401 and len(snaps) == len(want) == 8
    for (it, mu_g, M2_g, cov_g), (it_o, mu_o, M2_o, cov_o) in zip(snaps, want):
        assert it == it_o
        for name, x, y in (("mu", mu_g, mu_o), ("M2", M2_g, M2_o), ("cov", cov_g, cov_o)):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), (name, it)
    assert np.array_equal(end["AM"], AM)
    assert np.array_equal(end["nswap"], nswap) and end["swaps"] == nsw == 40 and nswap.sum() > 0
    for r in range(nt):                                       # 401 is no swap iteration: every rank holds its file's last row
        assert np.array_equal(end["X"][r], rows[r][200, :d]) and end["lnL"][r] == rows[r][200, -3]
        assert abs(end["nacc"][r] - last * rows[r][200, -2]) <= 1.0
    # the files: old rows untouched, 100 new rows each
    for f in names:
        new = open(tmp_path / f).read()
        assert new.startswith(old[f]) and len(new.splitlines()) == 301, f


def test_neff_stops_the_run_early(tmp_path, capsys):
    """neff (PTMCMCSampler.py:510-521): every 1000 iterations past 2 * burn the effective sample size of the cold chain
    is estimated (the restated acor, ess.acor) and the run ends once it reaches the request; the partial block is written.
    The stop is checked against an INDEPENDENT estimator on the same chain (Sokal window, ess.integrated_time; both are pinned
    by AR(1) series of known tau in tests/test_ess.py) and against the analytic rate of this sampler on this target."""
    from ptmcmcsampler_amd import PTSampler
    from ptmcmcsampler_amd.ess import AcorError, acor, integrated_time

    def taus(chain):                                                             # per dimension; None when acor gives up on one
        try:
            return [acor(chain[:, i])[0] for i in range(chain.shape[1])]
        except AcorError:
            return None
    d = 3
    s = PTSampler(d, ("iso",), ("flat",), np.eye(d), outDir=str(tmp_path), verbose=True, seed=2)
    s.sample(np.zeros(d), 200000, burn=300, thin=1, covUpdate=300, isave=1000, neff=150)
    out = capsys.readouterr().out
    assert s.Niter < 200000 and s.Niter % 1000 == 0 and s.Niter > 600
    m = re.search(r"Run Complete with (\d+) effective samples", out)
    assert m and int(m.group(1)) >= 150
    # the reference's expression, recomputed: iter / max(1, max_i acor(chain[burn:iter-1, i])[0])
    assert taus(s._chain[300:s.Niter - 1]) is not None                          # at the stop acor had an answer for every dimension
    tau_acor = np.nanmax(taus(s._chain[300:s.Niter - 1]))                       # nanmax, as the reference
    assert int(s.Niter / max(1.0, tau_acor)) == int(m.group(1))
    prev = s.Niter - 1000
    if prev > 600:                                                                   # it did not pass the test one check earlier
        tp = taus(s._chain[300:prev - 1])
        assert tp is None or int(prev / max(1.0, np.nanmax(tp))) < 150                # no estimate for a dimension = no stop
    # an independent estimator on the same samples agrees on the autocorrelation time within the sampling error of either
    # (a window of ~100 tau: relative standard error ~0.5 each; the estimators themselves are pinned on AR(1) series)
    tau_sokal = max(integrated_time(s._chain[300:s.Niter - 1, i]) for i in range(d))
    assert 0.4 < tau_acor / tau_sokal < 2.5, (tau_acor, tau_sokal)
    assert len(open(tmp_path / "chain_1.txt").read().splitlines()) == s.Niter + 1
    # without neff the same run goes on to the end
    s2 = PTSampler(d, ("iso",), ("flat",), np.eye(d), outDir=str(tmp_path / "b"), verbose=False, seed=2)
    s2.sample(np.zeros(d), 3000, burn=300, thin=1, covUpdate=300, isave=1000)
    assert s2.Niter == 3000


def test_continue_with_i0(tmp_path):
    """sample(p0, Niter, i0 = k), k != 0 (PTMCMCSampler.py:443-491): no re-initialisation; the chains take p0 as their
    state at iteration k (AM row k % covUpdate, sample k / thin) and run on to Niter."""
    from ptmcmcsampler_amd import PTSampler
    d = 4
    s = PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.1, outDir=str(tmp_path), verbose=False, seed=6, ntemps=2)
    with pytest.raises(RuntimeError, match="i0 = 0 first"):
        s.sample(np.zeros(d), 100, i0=50)
    kw = dict(burn=100, thin=5, covUpdate=40, isave=100, Tskip=10, maxIter=400)
    s.sample(np.zeros(d), 200, **kw)
    assert s.engine.iter == 200 and s.ind_next_write == 41
    p1 = np.full(d, 0.25)
    s.sample(p1, 400, i0=200, **kw)
    assert s.engine.iter == 400 and s.ind_next_write == 81
    assert np.array_equal(s._chain[40], p1) and np.isclose(s._lnlike[40], -0.5 * (p1 ** 2).sum())
    assert not np.array_equal(s._chain[41], p1) or not np.array_equal(s._chain[45], p1)
    assert len(open(tmp_path / "chain_1.0.txt").read().splitlines()) == 81


def test_config1_examples_simple_parameters(tmp_path, golden):
    """BASELINE configs[0] at exactly the parameters of the reference's examples/simple.py:52-122: 20-d dense Gaussian,
    box prior [0, 10], cov0 = 0.01 I, UniformJump with weight 5, sample(p0, 10000, burn=500, thin=1, covUpdate=500,
    SCAM = AM = DE = 20), Python callbacks.  The reference's own run on the same target (fixture config1.npz, written by
    tests/golden/make_golden.py) is the yardstick: same files, same cycle fractions, and two chains that agree within their
    Monte-Carlo error (different RNGs, so no closer)."""
    from ptmcmcsampler_amd import PTSampler
    g = golden("config1")
    ndim, pmin, pmax = int(g["ndim"]), float(g["pmin"]), float(g["pmax"])
    mu, icov = g["mu"], g["icov"]

    def lnlikefn(x):
        diff = x - mu
        return -np.dot(diff, np.dot(icov, diff)) / 2.0

    def lnpriorfn(x):
        return 0.0 if np.all(pmin <= x) and np.all(pmax >= x) else -np.inf

    s = PTSampler(ndim, lnlikefn, lnpriorfn, np.copy(g["cov0"]), outDir=str(tmp_path), verbose=False, seed=42)
    rs = np.random.RandomState(7)

    def jump(x, it, beta):
        return rs.uniform(pmin, pmax, len(x)), 0

    s.addProposalToCycle(jump, 5)
    s.sample(np.copy(g["p0"]), 10000, burn=500, thin=1, covUpdate=500, SCAMweight=20, AMweight=20, DEweight=20)
    assert str(g["chainfile_name"]) == "chain_1.txt"
    rows = open(tmp_path / "chain_1.txt").read().splitlines()
    assert len(rows) == int(g["nrows"]) == 10001
    assert sorted(open(tmp_path / "jumps.txt").read().splitlines()) == sorted(str(l) for l in g["jumps_txt"])
    assert sorted(s.jumpDict) == [str(n) for n in g["jnames"]]
    ref = {str(n): v for n, v in zip(g["jnames"], g["jstats"])}
    for name, (prop, acc) in s.jumpDict.items():
        assert abs(prop - ref[name][0]) < 5 * np.sqrt(ref[name][0]), name          # cycle shares (binomial error)
    assert sum(v[0] for v in s.jumpDict.values()) == 10000
    # 10000 iterations do not equilibrate this target: four runs of the reference itself (seeds 42..45 in the fixture)
    # differ by several posterior standard deviations.  Ours must sit inside their spread.
    accs, means, lls = g["ref_accs"], g["ref_means"], g["ref_lnlike_means"]
    assert accs.min() - 0.05 < s.naccepted / 10000.0 < accs.max() + 0.05
    x = s._chain[2500:10001]
    sd = np.sqrt(np.diag(g["ref_covs"].mean(0)))
    spread = max(np.max(np.abs(means[i] - means[j]) / sd) for i in range(4) for j in range(i))
    assert 2.0 < spread < 10.0                                                       # the yardstick itself (5.7 when generated)
    dist = min(np.max(np.abs(x.mean(0) - m) / sd) for m in means)
    assert dist < spread, (dist, spread)                                             # as close to one of them as they are to each other
    assert lls.min() - 15.0 < s._lnlike[2500:10001].mean() < 0.0
    assert np.all(x >= pmin) and np.all(x <= pmax)


def test_batched_device_callbacks_equal_the_fused_kernel(tmp_path):
    """batched=True: logl / logp are called once per iteration with the device tensor of all proposals (the reference's
    callback boundary, PTMCMCSampler.py:1072-1086 / :605-611, per batch).  With a callback that returns the bits of the
    built-in likelihood (2-d: one element per lane, so the sum has one order) the run equals the fused-kernel run."""
    import torch
    from ptmcmcsampler_amd import PTSampler
    d = 2
    kw = dict(burn=100, thin=1, covUpdate=50, isave=100, Tskip=10, SCAMweight=20, AMweight=20, DEweight=20)
    calls = []

    def logl(X, scale=1.0):
        assert X.is_cuda and X.dtype == torch.float64 and X.shape == (15, d)       # all chains of 5 walkers x 3 ranks at once
        calls.append(1)
        return -0.5 * scale * (X * X).sum(-1)

    def logp(X):
        return torch.where(((X >= -3.0) & (X <= 3.0)).all(-1), 0.0, -float("inf")).to(torch.float64)

    a = PTSampler(d, logl, logp, np.eye(d) * 0.5, loglkwargs=dict(scale=1.0), outDir=str(tmp_path / "a"), verbose=False, seed=4,
                  ntemps=3, nwalkers=5, keep_walkers=5, batched=True)
    a.sample(np.zeros(d), 300, **kw)
    b = PTSampler(d, ("iso",), ("box", -3.0 * np.ones(d), 3.0 * np.ones(d)), np.eye(d) * 0.5, outDir=str(tmp_path / "b"), verbose=False,
                  seed=4, ntemps=3, nwalkers=5, keep_walkers=5)
    b.sample(np.zeros(d), 300, **kw)
    assert len(calls) == 301                                                         # the start point + one call per iteration
    for name in ("X", "lnL", "lp", "slot_of", "nacc", "jstat", "nswap", "Ut"):
        assert np.array_equal(a.engine.get(name), b.engine.get(name)), name
    assert np.array_equal(a._chains, b._chains) and np.array_equal(a._lnlikes, b._lnlikes)
    assert (a.engine.get("jstat")[..., 0].sum(-1) > a.engine.get("jstat")[..., 1].sum(-1)).all()     # the prior rejected some
    assert open(tmp_path / "a" / "chain_1.0.txt").read() == open(tmp_path / "b" / "chain_1.0.txt").read()
    with pytest.raises(NotImplementedError, match="batched=True"):
        c = PTSampler(d, logl, logp, np.eye(d), outDir=str(tmp_path / "c"), verbose=False, batched=True)
        c.addProposalToCycle(lambda x, it, beta: (x, 0), 3)
        c.sample(np.zeros(d), 10)


def test_engine_modes_sample_the_same_posterior(tmp_path):
    """The engine modes that are not replicas of a reference run -- pooled covariance, one proposal-type draw per walker,
    odd/even swaps, covariance epochs factorized by the device Jacobi solver -- all on at once still sample the target:
    10-d dense Gaussian, 32 walkers x 4 temperatures, mean and covariance of the cold chains against the truth."""
    from ptmcmcsampler_amd import PTSampler
    d = 10
    rs = np.random.RandomState(12)
    A = rs.randn(d, d)
    C = A @ A.T / d + 0.3 * np.eye(d)
    mu = rs.randn(d)
    s = PTSampler(d, ("dense", mu, np.linalg.inv(C)), ("flat",), np.eye(d) * 0.01, outDir=str(tmp_path), verbose=False, seed=21,
                  ntemps=4, nwalkers=32, keep_walkers=32, cov_mode="pooled", pick_mode="walker", swap_mode="oddeven", eig_mode="jacobi")
    s.sample(mu + 0.1, 20000, burn=2000, thin=10, covUpdate=1000, isave=1000, Tskip=100)
    x = s._chains[:, 300:, :].reshape(-1, d)
    se = np.sqrt(np.diag(C) / (x.shape[0] / 40.0))
    assert np.all(np.abs(x.mean(0) - mu) < 5 * se)
    assert np.max(np.abs(np.cov(x.T) - C)) / np.max(np.abs(C)) < 0.12
    js = s.engine.get("jstat").astype(np.int64)[..., :3, 0]
    assert (js == js[:, :1]).all() and (js.sum(-1) == 20000).all() and (js.min() > 4000)      # uniform per walker, all three used
    assert s.engine.eig_epochs == 19 and s.nswap_accepted > 0
    # the adapted pooled covariance is the target's, up to the 2.4^2/d scaling being applied at proposal time, not here
    assert np.max(np.abs(s.cov - C)) / np.max(np.abs(C)) < 0.1


@pytest.mark.parametrize("cov_mode", ["per_walker", "pooled"])
def test_resume_a_batch_of_walkers_from_their_chain_files(tmp_path, capsys, cov_mode):
    """PTMCMCSampler.py:290-319, 591-599 for a BATCH of walkers at one temperature: every walker is a run of its own with a chain file
    of its own (chain_1.txt, chain_1_w<k>.txt with keep_walkers = nwalkers), and resume=True without a device checkpoint replays them
    side by side: the AM rings are the files' rows, the covariance epochs the oracle's Welford over each walker's rows (or the
    oracle's pooled statistics over all of them), the states the files' last rows; then sampling continues and every file grows."""
    from oracle import oracle as orc
    from ptmcmcsampler_amd import PTSampler
    d, W = 4, 3
    kw = dict(burn=200, thin=2, covUpdate=50, isave=100)
    p0 = np.full(d, 0.1)

    def make(resume):
        return PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path), verbose=False, seed=33, nwalkers=W, keep_walkers=W,
                         resume=resume, checkpoint=False, cov_mode=cov_mode)
    a = make(False)
    a.sample(p0, 400, **kw)
    assert not os.path.exists(tmp_path / "ptmi_checkpoint.npz")
    names = ["chain_1.txt"] + ["chain_1_w%d.txt" % k for k in range(1, W)]
    old = {f: open(tmp_path / f).read() for f in names}
    rows = [np.loadtxt(tmp_path / f) for f in names]
    assert all(r.shape == (201, d + 4) for r in rows) and not np.array_equal(rows[0], rows[1])
    b = make(True)
    snaps, end = [], {}
    replay = b._replay_chain_file

    def watched():
        eng = b.engine
        upd = eng.update_cov

        def update_cov(it_done):
            upd(it_done)
            snaps.append((it_done, eng.get("mu").copy(), eng.get("M2").copy(), eng.get("cov").copy()))

        eng.update_cov = update_cov
        last = replay()
        eng.update_cov = upd
        eng.sync()
        end.update(last=last, AM=eng.get("AM").copy(), X=eng.get("X").copy(), nacc=eng.get("nacc").astype(np.int64).copy())
        return last

    b._replay_chain_file = watched
    b.sample(p0, 600, **kw)
    assert "Resuming with 201 samples from file representing 401 original samples" in capsys.readouterr().out
    thin, cu, last = kw["thin"], kw["covUpdate"], 201 * kw["thin"] - 1
    Wc = W if cov_mode == "per_walker" else 1
    mu, M2, AM = np.zeros((Wc, d)), np.zeros((Wc, d, d)), np.zeros((W, cu, d))
    AM[:, 0] = [r[0, :d] for r in rows]
    want = []
    for it in range(1, last + 1):
        if (it - 1) % cu == 0 and it - 1 != 0:
            if cov_mode == "per_walker":
                cov = np.stack([orc.welford(AM[w], mu[w], M2[w], it - 1) for w in range(W)])
            elif b.engine.am_rle:                                    # replayed rows are stored rows (KEY flags): run lengths of one
                cov = orc.pool_update_rle(AM, np.full((W, cu), 2, dtype=np.uint64), mu[0], M2[0], it - 1)[None]
            else:
                cov = orc.pool_update(AM, mu[0], M2[0], it - 1)[None]
            want.append((it - 1, mu.copy(), M2.copy(), cov.copy()))
        AM[:, it % cu] = [r[it // thin, :d] for r in rows]
    assert end["last"] == last == 401 and len(snaps) == len(want) == 8
    for (i0, m0, q0, c0), (i1, m1, q1, c1) in zip(snaps, want):
        assert i0 == i1 and np.array_equal(m0, m1) and np.array_equal(q0, q1) and np.array_equal(c0, c1), i0
    assert np.array_equal(end["AM"], AM)
    assert np.array_equal(end["X"][:, 0], np.stack([r[200, :d] for r in rows]))
    assert np.array_equal(end["nacc"][:, 0], [int(round(last * r[200, -2])) for r in rows])
    for f in names:                                                  # the old rows stand, the new ones follow
        now = open(tmp_path / f).read()
        assert now.startswith(old[f]) and len(now.splitlines()) == 301
    # a ladder of several walkers still needs the checkpoint
    c = PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path / "c"), verbose=False, seed=1, nwalkers=2, keep_walkers=2, ntemps=2)
    c.sample(p0, 200, Tskip=10, **kw)
    c2 = PTSampler(d, ("iso",), ("flat",), np.eye(d) * 0.05, outDir=str(tmp_path / "c"), verbose=False, seed=1, nwalkers=2, keep_walkers=2, ntemps=2, resume=True)
    with pytest.raises(Exception, match="Couldn't resume"):
        c2.sample(p0, 400, Tskip=10, **kw)
