"""Integrated autocorrelation time and effective sample size.

The reference stops on ``neff`` with the third-party ``acor`` package (``acor.acor(chain[burn:iter-1, ii])[0]`` per
dimension, ``Neff = iter / max(1, nanmax(tau))``: PTMCMCSampler/PTMCMCSampler.py:510-521).  ``acor`` is not vendored
in the reference and not installable here (unpinned git master, README.md:155), so its PUBLISHED algorithm is restated
below (``acor``: J. Goodman's estimator as distributed in acor.cpp -- lag window of 10, recursive pairwise reduction of
the series) and the stop rule calls that.  No fixture of the reference pins it: parity "unpinned"; what pins it here is
the analytic autocorrelation time of AR(1) series (tests/test_ess.py).

``integrated_time`` / ``ess`` are the estimator bench.py reports ESS/sec with: FFT autocovariance and Sokal's adaptive
window (first M with M >= c tau(M)).  An estimate from fewer than ``MIN_TAUS`` autocorrelation times is flagged
(``reliable`` False) -- the window cannot have seen the slow modes then -- and refused under ``strict=True``.
"""
import numpy as np

MIN_TAUS = 50            # samples per autocorrelation time below which an estimate is not trusted (Sokal's rule of thumb)

# acor.cpp's constants
_TAUMAX, _WINMULT, _MINFAC = 2, 5, 5
_MAXLAG = _TAUMAX * _WINMULT


class AcorError(RuntimeError):
    """acor's own failure: "The autocorrelation time is too long relative to the variance"."""


def _acor_rec(x):
    """(sigma, tau) of a mean-free series by acor's recursion; x is overwritten."""
    L = len(x)
    if L < _MINFAC * _MAXLAG:
        raise AcorError("The autocorrelation time is too long relative to the variance")
    imax = L - _MAXLAG
    C = np.array([np.dot(x[:imax], x[s:s + imax]) for s in range(_MAXLAG + 1)]) / imax      # autocovariance at lags 0..MAXLAG
    D = C[0] + 2.0 * C[1:].sum()                                                              # diffusion coefficient
    with np.errstate(invalid="ignore"):
        sigma = np.sqrt(D / L)                 # NaN for a negative estimate, as the C code
    tau = D / C[0] if C[0] > 0 else np.nan
    if not tau * _WINMULT >= _MAXLAG:          # the lag window covers WINMULT autocorrelation times: done (NaN ends here too)
        return sigma, tau
    Lh = L // 2                                # else: sum neighbours (the same diffusion coefficient at half the length) and recurse
    y = x[0:2 * Lh:2] + x[1:2 * Lh:2]
    y -= y.mean()
    sigma, _ = _acor_rec(y)
    D = 0.25 * sigma * sigma * L
    return np.sqrt(D / L), D / C[0]


def acor(x):
    """``(tau, mean, sigma)`` as ``acor.acor(x)`` returns them: integrated autocorrelation time, mean, and the standard
    error of the mean.  Raises AcorError when the series is too short for its autocorrelation time."""
    x = np.array(x, dtype=np.float64).ravel()
    mean = x.mean() if len(x) else 0.0
    sigma, tau = _acor_rec(x - mean)
    return float(tau), float(mean), float(sigma)


def autocorrelation(x):
    """Normalized autocorrelation function along axis 0 of ``x`` ([N] or [N][k]) by FFT, lags 0 .. N-1."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    x = x - x.mean(axis=0)
    nfft = 1 << (2 * n - 1).bit_length()
    f = np.fft.rfft(x, nfft, axis=0)
    acf = np.fft.irfft(f * np.conjugate(f), nfft, axis=0)[:n]
    c0 = acf[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        return acf / c0


def integrated_time(x, c=5.0, full=False):
    """Integrated autocorrelation time(s) of the series along axis 0 of ``x`` ([N] -> float, [N][k] -> array[k]):
    tau(M) = 1 + 2 sum_{s<=M} rho(s) at Sokal's window, the first M with M >= c tau(M).  A constant series has tau = 1.
    ``full=True`` returns a dict with ``tau``, ``window`` and ``reliable`` (N >= MIN_TAUS tau and a window was found)."""
    x = np.asarray(x, dtype=np.float64)
    one = x.ndim == 1
    if one:
        x = x[:, None]
    n, k = x.shape
    if n < 4:
        tau, win, ok = np.ones(k), np.zeros(k, dtype=np.int64), np.zeros(k, dtype=bool)
    else:
        rho = autocorrelation(x)
        rho[:, ~np.isfinite(rho[0])] = 0.0                         # constant columns
        taus = 2.0 * np.cumsum(rho, axis=0) - 1.0
        hit = np.arange(n)[:, None] >= c * taus
        found = hit.any(axis=0)
        win = np.where(found, hit.argmax(axis=0), n - 1)
        tau = np.maximum(taus[win, np.arange(k)], 1.0)
        ok = found & (n >= MIN_TAUS * tau)
    if full:
        return dict(tau=float(tau[0]) if one else tau, window=int(win[0]) if one else win, reliable=bool(ok[0]) if one else ok, n=n)
    return float(tau[0]) if one else tau


def ess(chain, c=5.0, strict=False):
    """min over dimensions of N / tau for a [N][d] chain.  ``strict``: raise ValueError when the chain is shorter than
    MIN_TAUS autocorrelation times in some dimension (the estimate would be a guess)."""
    chain = np.asarray(chain, dtype=np.float64)
    if chain.ndim == 1:
        chain = chain[:, None]
    r = integrated_time(chain, c, full=True)
    if strict and not np.all(r["reliable"]):
        raise ValueError("chain of %d samples is shorter than %d autocorrelation times (tau up to %.1f): ESS not estimable"
                         % (chain.shape[0], MIN_TAUS, float(np.max(r["tau"]))))
    return float(chain.shape[0] / np.max(r["tau"]))
