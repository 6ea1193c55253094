"""The engine's device eigensolver (eig_mode="jacobi", include/ptmi.h ptmi_eig_jacobi) replaces np.linalg.svd of the
adapted covariance (PTMCMCSampler.py:797-803).  CPU: the oracle's restatement of the algorithm against LAPACK.  GPU: the
kernel against the oracle, bit for bit, alone and inside a sampling run."""
import numpy as np
import pytest

from oracle import oracle as orc


def _spd(d, rs, floor=0.1):
    A = rs.randn(d, d)
    c = A @ A.T / d + floor * np.eye(d)
    return (c + c.T) / 2


@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 33, 99, 100])
def test_oracle_jacobi_factorizes_like_lapack(d):
    rs = np.random.RandomState(d)
    cov = _spd(d, rs) * 10.0 ** rs.uniform(-6, 3)
    Ut, S, sweeps = orc.eig_jacobi(cov)
    assert 0 < sweeps + (d == 1) < orc.JACOBI_MAX_SWEEPS
    scale = np.abs(cov).max()
    assert np.abs(Ut.T @ np.diag(S) @ Ut - cov).max() <= 1e-12 * scale        # U diag(S) U^T = cov
    assert np.abs(Ut @ Ut.T - np.eye(d)).max() <= 1e-12
    assert (np.diff(S) <= 0).all() and (S >= 0).all()
    w = np.linalg.svd(cov, compute_uv=False)                                    # what the reference's call returns
    assert np.abs(S - w).max() <= 1e-12 * w.max()
    big = np.abs(Ut).argmax(axis=1)
    assert (Ut[np.arange(d), big] > 0).all()                                    # the sign rule
    # same subspaces as LAPACK: |cos| of matching eigenvectors is 1 (the spectrum of a random SPD matrix is simple)
    U, _, _ = np.linalg.svd(cov)
    assert np.abs(np.abs(np.einsum("ki,ik->k", Ut, U)) - 1).max() <= 1e-8


def test_oracle_jacobi_degenerate_inputs():
    Ut, S, n = orc.eig_jacobi(np.eye(6) * 0.01)               # the sampler's usual start: nothing to rotate
    assert n == 0 and np.array_equal(Ut, np.eye(6)) and np.array_equal(S, np.full(6, 0.01))
    Ut, S, n = orc.eig_jacobi(np.zeros((4, 4)))
    assert np.array_equal(Ut, np.eye(4)) and not S.any()
    rs = np.random.RandomState(1)
    cov = _spd(7, rs)
    cov[:, 2] = 0
    cov[2, :] = 0                                              # a parameter that never moved: one zero eigenvalue
    Ut, S, _ = orc.eig_jacobi(cov)
    assert S[-1] == 0.0 and abs(abs(Ut[-1, 2]) - 1) < 1e-15
    assert np.abs(Ut.T @ np.diag(S) @ Ut - cov).max() <= 1e-13
    v = rs.randn(5)
    Ut, S, _ = orc.eig_jacobi(np.outer(v, v))                  # rank one
    assert abs(S[0] - v @ v) <= 1e-13 * (v @ v) and np.abs(S[1:]).max() <= 1e-13 * (v @ v)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2, 7, 50, 99, 100, 101])
def test_device_jacobi_is_bit_identical_to_the_oracle(d):
    from ptmcmcsampler_amd import _lib
    from ptmcmcsampler_amd.engine import PTEngine
    rs = np.random.RandomState(100 + d)
    W = 5
    g = PTEngine(d, 1, W, np.eye(d), weights=(1, 0, 0), cov_update=4, burn=4, tskip=0, eig_mode="jacobi")
    covs = np.stack([_spd(d, rs) * 10.0 ** rs.uniform(-4, 2) for _ in range(W - 2)] + [np.eye(d) * 0.01, np.zeros((d, d))])
    g.put("cov", covs)
    _lib.check(g.lib.ptmi_eig_jacobi(g.h))
    g.sync()
    Ut, S = g.get("Ut")[:, 0], g.get("S")[:, 0]
    for w in range(W):
        oUt, oS, _ = orc.eig_jacobi(covs[w])
        assert np.array_equal(Ut[w].view(np.uint64), oUt.view(np.uint64)), (d, w)
        assert np.array_equal(S[w].view(np.uint64), oS.view(np.uint64)), (d, w)


@pytest.mark.parametrize("d", [1, 2, 3, 5, 8, 33, 99, 100, 101, 128])
def test_oracle_ql_factorizes_like_lapack(d):
    """orc_eig_ql (Householder tridiagonalization + implicit QL; the engine's eig_mode "ql") against LAPACK."""
    rs = np.random.RandomState(d)
    cov = _spd(d, rs) * 10.0 ** rs.uniform(-6, 3)
    Ut, S, iters = orc.eig_ql(cov)
    scale = np.abs(cov).max()
    assert np.abs(Ut.T @ np.diag(S) @ Ut - cov).max() <= 1e-12 * scale
    assert np.abs(Ut @ Ut.T - np.eye(d)).max() <= 1e-12
    assert (np.diff(S) <= 0).all() and (S >= 0).all()
    w = np.linalg.svd(cov, compute_uv=False)
    assert np.abs(S - w).max() <= 1e-12 * w.max()
    big = np.abs(Ut).argmax(axis=1)
    assert (Ut[np.arange(d), big] > 0).all()
    U, _, _ = np.linalg.svd(cov)
    assert np.abs(np.abs(np.einsum("ki,ik->k", Ut, U)) - 1).max() <= 1e-8
    # a nearly degenerate spectrum (the cumulative covariance of an isotropic target): where the Jacobi sweeps are slow
    X = rs.randn(4000, d)
    C = np.cov(X.T) if d > 1 else np.array([[1.0]])
    Ut, S, iters = orc.eig_ql(C)
    assert np.abs(Ut.T @ np.diag(S) @ Ut - C).max() <= 1e-12 * np.abs(C).max() and iters <= 3 * d + 3


def test_oracle_ql_degenerate_inputs():
    Ut, S, n = orc.eig_ql(np.eye(6) * 0.01)
    assert n == 0 and np.array_equal(np.abs(Ut), np.eye(6)) and np.array_equal(S, np.full(6, 0.01))
    Ut, S, n = orc.eig_ql(np.zeros((4, 4)))
    assert not S.any() and np.abs(Ut @ Ut.T - np.eye(4)).max() == 0
    rs = np.random.RandomState(1)
    cov = _spd(7, rs)
    cov[:, 2] = 0
    cov[2, :] = 0
    Ut, S, _ = orc.eig_ql(cov)
    assert S[-1] <= 1e-15 and abs(abs(Ut[-1, 2]) - 1) < 1e-12
    assert np.abs(Ut.T @ np.diag(S) @ Ut - cov).max() <= 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("d", [1, 2, 7, 50, 99, 100, 101, 128])
def test_device_ql_is_bit_identical_to_the_oracle(d, split, monkeypatch):
    """ptmi_eig_ql in both forms: one kernel per matrix (few matrices), and reduce -> the scalar chains of all matrices at once ->
    apply (many matrices: per-walker covariances)."""
    monkeypatch.setenv("PTMI_QL_SPLIT", split)
    from ptmcmcsampler_amd import _lib
    from ptmcmcsampler_amd.engine import PTEngine
    rs = np.random.RandomState(200 + d)
    W = 6
    g = PTEngine(d, 1, W, np.eye(d), weights=(1, 0, 0), cov_update=4, burn=4, tskip=0, eig_mode="ql")
    X = rs.randn(3000, d)
    covs = np.stack([_spd(d, rs) * 10.0 ** rs.uniform(-4, 2) for _ in range(W - 3)] +
                    [np.cov(X.T).reshape(d, d), np.eye(d) * 0.01, np.zeros((d, d))])
    g.put("cov", covs)
    _lib.check(g.lib.ptmi_eig_ql(g.h))
    g.sync()
    Ut, S = g.get("Ut")[:, 0], g.get("S")[:, 0]
    for w in range(W):
        oUt, oS, _ = orc.eig_ql(covs[w])
        assert np.array_equal(Ut[w].view(np.uint64), oUt.view(np.uint64)), (d, w)
        assert np.array_equal(S[w].view(np.uint64), oS.view(np.uint64)), (d, w)


@pytest.mark.gpu
@pytest.mark.parametrize("cov_mode,d,nt,W,eig", [("per_walker", 100, 4, 6, "jacobi"), ("per_walker", 12, 64, 3, "jacobi"), ("pooled", 100, 3, 40, "jacobi"),
                                                 ("per_walker", 100, 4, 6, "ql"), ("per_walker", 12, 64, 3, "ql"), ("pooled", 100, 3, 40, "ql")])
def test_sampling_with_device_eigensolver_matches_oracle(cov_mode, d, nt, W, eig):
    """A whole run adapted through ptmi_eig_jacobi / ptmi_eig_ql: every array equals the oracle's run adapted through orc_eig_jacobi / orc_eig_ql."""
    from ptmcmcsampler_amd.engine import PTEngine
    kw = dict(weights=(20, 20, 20), cov_update=40, burn=80, tskip=10, seed=5, cov_mode=cov_mode, eig_mode=eig)
    rs = np.random.RandomState(3)
    cov0, p0 = _spd(d, rs) * 0.01, rs.randn(W, nt, d) * 0.3
    g, o = PTEngine(d, nt, W, cov0, **kw), orc.OracleEngine(d, nt, W, cov0, **kw)
    for e in (g, o):
        e.init_state(p0)
        e.run(170)
    g.sync()
    for name in ("X", "lnL", "slot_of", "nacc", "jstat", "nswap", "cov", "Ut", "S") + (("AM",) if cov_mode == "per_walker" else ()):
        a, b = g.get(name), getattr(o, name)
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), name
    assert g.eig_epochs == 4 and o.jstat[..., 1, 1].sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("cov_mode,d,nt,W", [("pooled", 300, 4, 12), ("per_walker", 40, 3, 5)])
def test_library_eigensolver_on_the_stream(cov_mode, d, nt, W):
    """eig_mode="hipsolver": the ROCm library's symmetric eigensolver factorizes the adapted covariance on the engine's
    stream (the large-ndim choice; PTMCMCSampler.py:797-803 calls LAPACK).  Its last bits are the library's, so what is
    checked is the decomposition the proposals use -- U diag(S) U^T = cov, orthonormal rows, eigenvalues descending -- and
    that the chains stay exact samples of their own arithmetic (lnL of the held row, proposal counters)."""
    from ptmcmcsampler_amd.engine import PTEngine
    rs = np.random.RandomState(8)
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 20, 0), cov_update=50, burn=1000, tskip=10, seed=3, cov_mode=cov_mode,
                 eig_mode="hipsolver")
    g.init_state(rs.randn(W, nt, d) * 0.2)
    g.run(130)
    g.sync()
    assert g.eig_epochs == 2
    cov, Ut, S = g.get("cov"), g.get("Ut")[:, 0], g.get("S")[:, 0]
    for w in range(g.Wc):
        assert np.allclose(Ut[w].T @ np.diag(S[w]) @ Ut[w], cov[w], rtol=0, atol=1e-12 * np.abs(cov[w]).max())
        assert np.allclose(Ut[w] @ Ut[w].T, np.eye(d), atol=1e-12)
        assert (np.diff(S[w]) <= 1e-15 * S[w].max()).all() and (S[w] >= 0).all()
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * (X ** 2).sum(-1), rtol=1e-12)
    js = g.get("jstat").astype(np.int64)
    assert (js[..., :2, 0].sum(-1) == 130).all() and js[..., 1, 1].sum() > 0


def _sytrd_matrix(kind, d):
    rng = np.random.default_rng(d)
    if kind == "scaled":                                          # a random ill-scaled covariance
        X = rng.standard_normal((4 * d, d)) * np.exp(rng.uniform(-2, 2, d))
        return X.T @ X / (4 * d)
    if kind == "isotropic":                                       # what an isotropic target adapts to: a nearly degenerate spectrum, 1 +- a few per cent
        X = rng.standard_normal((50 * d, d))
        return X.T @ X / (50 * d)
    if kind == "identity":                                        # the identity up to rounding noise: everything deflates
        E = rng.standard_normal((d, d)) * 1e-14
        return np.eye(d) + 0.5 * (E + E.T)
    if kind == "clusters":                                        # a few well-separated clusters of (nearly) equal eigenvalues
        Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        w = np.repeat([1e-3, 1.0, 1.0 + 1e-9, 50.0], (d + 3) // 4)[:d] * (1 + 1e-13 * rng.standard_normal(d))
        C = (Q * w) @ Q.T
        return 0.5 * (C + C.T)
    if kind == "wilkinson":                                       # tridiagonal already: |i - d/2| on the diagonal, ones beside it (pairs of close eigenvalues)
        return np.diag(np.abs(np.arange(d) - d // 2).astype(float)) + np.diag(np.ones(d - 1), 1) + np.diag(np.ones(d - 1), -1)
    raise ValueError(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("d,kind", [(3, "scaled"), (37, "scaled"), (130, "scaled"), (300, "scaled"), (1000, "scaled"), (33, "isotropic"), (65, "wilkinson"),
                                    (500, "isotropic"), (1000, "isotropic"), (1024, "identity"), (700, "clusters"), (777, "wilkinson"), (64, "identity")])
def test_sytrd_eigensolver_decomposes_a_covariance(d, kind):
    """ptmi_eig_sytrd (eig_mode="sytrd": Householder tridiagonalization in one kernel with the matrix in the LDS of its blocks, then the
    engine's divide-and-conquer solver of the tridiagonal matrix -- leaves by implicit QL, merges with deflation, secular roots by
    bisection, Gu-Eisenstat's recomputed weights, the vectors multiplied on the matrix cores -- and the back-transformation through
    the reflectors; csrc/ptmi_dc.inc.h): eigenvalues against numpy.linalg.eigvalsh, orthonormal rows, U diag(S) U^T = cov, on
    ill-scaled, nearly degenerate, clustered, near-identity and Wilkinson matrices.  Replaces np.linalg.svd of
    PTMCMCSampler.py:797-803 for one large pooled covariance."""
    import torch
    from ptmcmcsampler_amd import _lib
    from ptmcmcsampler_amd.engine import PTEngine
    g = PTEngine(d, 2, 2, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=100, burn=1000, tskip=10, seed=1, cov_mode="pooled",
                 use_de_buffer=False, eig_mode="sytrd")
    g.init_state(np.zeros(d))
    cov = _sytrd_matrix(kind, d)
    g.t["cov"][0].copy_(torch.from_numpy(cov))
    for _ in range(2):                                              # twice: the scratch and the barrier word are reused
        _lib.check(g.lib.ptmi_eig_sytrd(g.h, None, None, None))
    g.sync()
    Ut, S = g.get("Ut")[0, 0], g.get("S")[0, 0]
    ev = np.linalg.eigvalsh(cov)
    w = np.sort(np.abs(ev))[::-1]
    assert np.allclose(S, w, rtol=0, atol=1e-12 * w.max()), np.abs(S - w).max() / w.max()
    assert np.allclose(Ut @ Ut.T, np.eye(d), atol=1e-12), np.abs(Ut @ Ut.T - np.eye(d)).max()
    if ev.min() >= 0:                                             # (S holds absolute values: the reconstruction needs a definite matrix)
        assert np.allclose((Ut.T * S) @ Ut, cov, rtol=0, atol=1e-12 * np.abs(cov).max())
    assert np.abs(cov @ Ut.T - Ut.T * (np.sign(np.einsum("ki,ij,kj->k", Ut, cov, Ut)) * S)).max() <= 1e-12 * w.max()      # residual of every pair
    assert (np.diff(S) <= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("lag", [0, 3])
def test_sytrd_eigensolver_in_a_run(lag):
    """A pooled run adapted through eig_mode="sytrd", at once and on the side stream (eig_lag): the decomposition the proposals use
    and the chains' own arithmetic, as for the library's eigensolver above."""
    from ptmcmcsampler_amd.engine import PTEngine
    d, nt, W = 300, 4, 12
    rs = np.random.RandomState(8)
    g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 20, 0), cov_update=50, burn=1000, tskip=10, seed=3, cov_mode="pooled",
                 eig_mode="sytrd", eig_lag=lag)
    g.init_state(rs.randn(W, nt, d) * 0.2)
    g.run(130)
    g._eig_finish()
    g.sync()
    assert g.eig_epochs == 2
    cov, Ut, S = g.get("cov")[0], g.get("Ut")[0, 0], g.get("S")[0, 0]
    assert np.allclose(Ut.T @ np.diag(S) @ Ut, cov, rtol=0, atol=1e-12 * np.abs(cov).max())
    assert np.allclose(Ut @ Ut.T, np.eye(d), atol=1e-12)
    X, lnL = g.get("X"), g.get("lnL")
    assert np.allclose(lnL, -0.5 * (X ** 2).sum(-1), rtol=1e-12)
    js = g.get("jstat").astype(np.int64)
    assert (js[..., :2, 0].sum(-1) == 130).all() and js[..., 1, 1].sum() > 0
