"""The pooled covariance the engine adapts (cov_mode="pooled": slab-wise shifted sums of outer products, combined with the
running statistics by Chan's formula -- oracle/ptmcmc_oracle.c orc_pool_update, the definition the HIP kernels are held to
bit for bit) IS the sample covariance of all walkers' rank-0 samples -- the batched counterpart of the reference's cumulative
`_updateRecursive` estimate (PTMCMCSampler.py:769-794), which for one walker equals np.cov of that walker's samples."""
import numpy as np
import pytest

from oracle import oracle as orc


@pytest.mark.parametrize("d,W,mem,epochs,offset", [(7, 5, 40, 3, 3.0), (20, 130, 25, 2, 3.0), (100, 70, 30, 2, 3.0), (12, 700, 10, 3, 1e6),
                                                   (120, 9, 15, 2, -40.0)])
def test_pooled_covariance_is_the_sample_covariance_of_all_cold_samples(d, W, mem, epochs, offset):
    rs = np.random.RandomState(d + W)
    A = rs.randn(d, d)
    L = np.linalg.cholesky(A @ A.T / d + 0.3 * np.eye(d))
    mu, M2 = np.zeros(d), np.zeros((d, d))
    seen = []
    for ep in range(1, epochs + 1):
        AM = (rs.randn(W, mem, d) @ L.T) + rs.randn(d) * 0.1 + offset       # correlated rows around a (far) non-zero mean
        seen.append(AM)
        cov_o = orc.pool_update(AM, mu, M2, ep * mem)
        rows = np.concatenate([s.reshape(-1, d) for s in seen])
        # the yardstick in extended precision: the shifted sums must not lose the covariance under a mean 1e6 sigma away
        rl = rows.astype(np.longdouble)
        rc = rl - rl.mean(0)
        ref = np.asarray(rc.T @ rc / (len(rows) - 1), dtype=np.float64)
        assert np.max(np.abs(cov_o - ref)) <= 1e-9 * np.max(np.abs(ref)), ep
        assert np.max(np.abs(mu - rows.mean(0))) <= 1e-12 * np.max(np.abs(rows.mean(0)))
        assert np.array_equal(M2, M2.T) and np.array_equal(cov_o, cov_o.T)       # mirrored upper triangle
        # the slab size is part of the definition (summation order) but not of the value
        mu2, M22 = np.zeros(d), np.zeros((d, d))
        for e2, am in enumerate(seen, 1):
            cov2 = orc.pool_update(am, mu2, M22, e2 * mem, slab=3)
        assert np.max(np.abs(cov2 - cov_o)) <= 1e-12 * np.max(np.abs(cov_o))
    # one walker: the reference's own definition, np.cov of its rows (PTMCMCSampler.py:769-794)
    mu1, M21 = np.zeros(d), np.zeros((d, d))
    for e2, am in enumerate(seen, 1):
        c1 = orc.welford(am[0], mu1, M21, e2 * mem)
    own = np.concatenate([s[0] for s in seen])
    assert np.max(np.abs(c1 - np.cov(own, rowvar=False))) <= 1e-7 * np.max(np.abs(c1))


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
