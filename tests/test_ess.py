"""The ESS estimators against series of KNOWN integrated autocorrelation time (CPU).

An AR(1) process x_t = rho x_{t-1} + e_t has rho(s) = rho^s and tau = 1 + 2 sum rho^s = (1 + rho) / (1 - rho).
Pins ptmcmcsampler_amd/ess.py: ``integrated_time`` / ``ess`` (Sokal window; bench.py's ESS/sec) and ``acor`` (the restated
published algorithm of the un-vendored package the reference's neff stop calls, PTMCMCSampler.py:510-521)."""
import numpy as np
import pytest
from scipy.signal import lfilter

from ptmcmcsampler_amd import ess as E


def ar1(rho, n, seed, k=None):
    rs = np.random.RandomState(seed)
    e = rs.standard_normal((n,) if k is None else (n, k))
    x = lfilter([1.0], [1.0, -rho], e, axis=0)
    return x[n // 10:]                                   # drop the transient from x_0 = e_0


@pytest.mark.parametrize("rho,n", [(0.5, 1 << 18), (0.9, 1 << 20), (0.99, 1 << 22)])
def test_sokal_window_recovers_the_ar1_autocorrelation_time(rho, n):
    tau_true = (1 + rho) / (1 - rho)
    x = ar1(rho, n, seed=int(rho * 100))
    r = E.integrated_time(x, full=True)
    assert r["reliable"] and r["window"] >= 5 * r["tau"] - 1
    assert abs(r["tau"] / tau_true - 1) < 0.10, (r, tau_true)
    assert abs(E.ess(x, strict=True) / (len(x) / tau_true) - 1) < 0.10


@pytest.mark.parametrize("rho,n", [(0.5, 1 << 18), (0.9, 1 << 20), (0.99, 1 << 22)])
def test_acor_restatement_recovers_the_ar1_autocorrelation_time(rho, n):
    tau_true = (1 + rho) / (1 - rho)
    x = ar1(rho, n, seed=7 + int(rho * 100))
    tau, mean, sigma = E.acor(x)
    assert abs(tau / tau_true - 1) < 0.10, (tau, tau_true)
    assert abs(mean - x.mean()) < 1e-12
    # sigma is the standard error of the mean: sqrt(var * tau / N)
    assert abs(sigma / np.sqrt(x.var() * tau_true / len(x)) - 1) < 0.10
    # and the two estimators agree with each other on the same series
    assert abs(tau / E.integrated_time(x) - 1) < 0.10


def test_columns_are_independent_and_ess_takes_the_slowest():
    x = np.stack([ar1(0.5, 1 << 18, 1), ar1(0.9, 1 << 18, 2), np.random.RandomState(3).standard_normal((1 << 18) - (1 << 18) // 10)], 1)
    tau = E.integrated_time(x)
    assert abs(tau[0] / 3 - 1) < 0.1 and abs(tau[1] / 19 - 1) < 0.1 and abs(tau[2] - 1) < 0.05
    for j in range(3):
        assert tau[j] == E.integrated_time(x[:, j])                      # the vectorized form is the scalar one, column by column
    assert E.ess(x) == len(x) / tau.max()


def test_short_series_is_flagged_and_refused():
    """A window shorter than MIN_TAUS autocorrelation times cannot have seen the slow mode: flagged, refused under strict."""
    rho = 0.99                                                           # tau = 199
    x = ar1(rho, 5000, seed=5)                                           # 4500 samples = 23 tau
    r = E.integrated_time(x, full=True)
    assert not r["reliable"]
    with pytest.raises(ValueError):
        E.ess(x, strict=True)
    assert E.ess(x) > 0                                                  # the non-strict call still answers
    # a non-stationary series (a drift: the bench's old window started 500 iterations after p0 = 0) never finds a window
    t = np.arange(4000.0)
    r = E.integrated_time(t + np.random.RandomState(1).standard_normal(4000), full=True)
    assert not r["reliable"]
    # acor itself refuses a series shorter than MINFAC * MAXLAG = 50 samples, and one whose recursion runs out of samples
    with pytest.raises(E.AcorError):
        E.acor(np.random.RandomState(2).standard_normal(40))
    with pytest.raises(E.AcorError):
        E.acor(ar1(0.999, 3000, seed=9))


def test_degenerate_series():
    assert E.integrated_time(np.ones(100)) == 1.0
    assert E.integrated_time(np.arange(3.0)) == 1.0
    w = np.random.RandomState(0).standard_normal(1 << 16)
    assert abs(E.integrated_time(w) - 1) < 0.05 and abs(E.acor(w)[0] - 1) < 0.05
