"""The running mean's division in welford_rows_kernel (ptmi_abi.hip div_by_count) against the reference's plain `diff / n`
(PTMCMCSampler.py:787): reciprocal times dividend, corrected twice through the exact remainder, is the correctly rounded quotient
for every integer count -- random dividends, dividends next to rounding ties of the quotient, and tiny / huge magnitudes."""
import numpy as np
from oracle import oracle as O


def _cases(rng, n):
    cnt = np.concatenate([rng.integers(1, 100001, n // 2), rng.integers(1, 2_000_000_000, n - n // 2)]).astype(np.float64)
    a0 = rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))
    # quotients a hair away from the midpoint of two doubles: a = n * (m + half an ulp), nudged one ulp either way
    m = 1.0 + rng.integers(0, 2 ** 52, n).astype(np.float64) * 2.0 ** -52
    tie = np.nextafter(m * cnt, np.where(rng.integers(0, 2, n) == 1, np.inf, -np.inf))
    return np.concatenate([a0, tie, -tie]), np.concatenate([cnt, cnt, cnt])


def test_division_by_the_row_count_is_the_plain_division():
    rng = np.random.default_rng(5)
    a, n = _cases(rng, 2_000_000)
    q = O.div_by_count(a, n)
    assert np.array_equal(q.view(np.uint64), (a / n).view(np.uint64))


def test_small_and_special_dividends():
    n = np.array([1.0, 3.0, 7.0, 1000.0, 999983.0] * 4)
    a = np.array([0.0] * 5 + [-0.0] * 5 + [5e-324] * 5 + [1e-305] * 5)
    q = O.div_by_count(a, n)
    assert np.array_equal(q.view(np.uint64), (a / n).view(np.uint64))
