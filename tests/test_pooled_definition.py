"""The pooled covariance the engine adapts (cov_mode="pooled": per-walker fused Welford + two-level Chan combination,
oracle/ptmcmc_oracle.c orc_welford2(fused=1) + orc_pool_cov, the definition the HIP kernels are held to bit for bit) IS the
sample covariance of all walkers' rank-0 samples -- the batched counterpart of the reference's cumulative `_updateRecursive`
estimate (PTMCMCSampler.py:769-794), which for one walker equals np.cov of that walker's samples."""
import numpy as np
import pytest

from oracle import oracle as orc


@pytest.mark.parametrize("d,W,mem,epochs", [(7, 5, 40, 3), (20, 130, 25, 2), (100, 70, 30, 2)])
def test_pooled_covariance_is_the_sample_covariance_of_all_cold_samples(d, W, mem, epochs):
    rs = np.random.RandomState(d + W)
    A = rs.randn(d, d)
    L = np.linalg.cholesky(A @ A.T / d + 0.3 * np.eye(d))
    mu, M2 = np.zeros((W, d)), np.zeros((W, d, d))
    seen = []
    for ep in range(1, epochs + 1):
        AM = (rs.randn(W, mem, d) @ L.T) + rs.randn(d) * 0.1 + 3.0          # correlated rows around a non-zero mean
        seen.append(AM)
        for w in range(W):
            c = orc.welford(AM[w], mu[w], M2[w], ep * mem, fused=True)
            # one walker's running estimate is the sample covariance of its own rows (the reference's definition)
            own = np.concatenate([s[w] for s in seen])
            assert np.max(np.abs(c - np.cov(own, rowvar=False))) <= 1e-10 * np.max(np.abs(c))
            assert np.array_equal(M2[w], M2[w].T)                              # mirrored upper triangle
        mu_o, cov_o = np.zeros(d), np.zeros((d, d))
        orc.lib().orc_pool_cov(d, W, ep * mem, orc._p(mu), orc._p(M2), orc._p(mu_o), orc._p(cov_o))
        rows = np.concatenate([s.reshape(-1, d) for s in seen])
        ref = np.cov(rows, rowvar=False)
        assert np.max(np.abs(cov_o - ref)) <= 1e-10 * np.max(np.abs(ref)), ep
        assert np.max(np.abs(mu_o - rows.mean(0))) <= 1e-12 * np.max(np.abs(rows.mean(0)))


def test_reference_welford_equals_numpy_cov_on_the_reference_fixture(golden):
    """Same statement for the unfused (reference) arithmetic on the reference's own buffers: cov after epoch e is
    np.cov of the rows seen so far in buffer order (tests/golden/welford.npz, PTMCMCSampler.py:778-794)."""
    g = golden("welford")
    for d in (5, 100):
        rows = []
        for ep in range(3):
            rows.append(g["am_d%d_e%d" % (d, ep)])
            ref = np.cov(np.concatenate(rows), rowvar=False)
            cov = g["cov_d%d_e%d" % (d, ep)]
            assert np.max(np.abs(cov - ref)) <= 1e-10 * np.max(np.abs(ref))
