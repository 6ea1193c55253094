"""Effective sample size with Sokal's adaptive window (our own estimator).

The reference stops on ``neff`` using the un-vendored third-party ``acor`` package
(PTMCMCSampler/PTMCMCSampler.py:510-521); no reference test pins its value, so parity
for this quantity is "unpinned".  The same code is applied to every chain we compare."""
import numpy as np


def integrated_time(x, c=5.0):
    """Integrated autocorrelation time of a 1-d series (FFT autocovariance, window M >= c*tau)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    if n < 4:
        return 1.0
    x = x - x.mean()
    nfft = 1 << (2 * n - 1).bit_length()
    f = np.fft.rfft(x, nfft)
    acf = np.fft.irfft(f * np.conjugate(f), nfft)[:n]
    if acf[0] <= 0:
        return 1.0
    acf = acf / acf[0]
    tau = 2.0 * np.cumsum(acf) - 1.0
    m = np.arange(n) >= c * tau
    win = int(np.argmax(m)) if m.any() else n - 1
    return float(max(tau[win], 1.0))


def ess(chain, c=5.0):
    """min over dimensions of N / tau for a [N][d] chain."""
    chain = np.asarray(chain, dtype=np.float64)
    if chain.ndim == 1:
        chain = chain[:, None]
    n = chain.shape[0]
    return min(n / integrated_time(chain[:, j], c) for j in range(chain.shape[1]))
