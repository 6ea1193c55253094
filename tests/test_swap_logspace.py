"""The swap test in log space against the reference's form, on adversarial near-ties (CPU; no GPU, no /root/reference).

The reference accepts pair k iff ``u <= exp(sum)`` (PTMCMCSampler.py:678-679, numpy's libm ``exp``); the oracle and the HIP
sweep test ``log u <= sum`` with their own correctly-ordered ``log`` (oracle/ptmcmc_oracle.c orc_log == csrc/ptmi_device.h
det_log, bit for bit on the GPU), because the logarithm depends on the uniform alone and leaves the sweep's pair-to-pair
recurrence.  Both forms are monotone in u, so they can only differ where rounding moves one of them across the tie: this
test builds 10^7 pairs (u, sum) with u within a few ulp of exp(sum) -- the ONLY place a disagreement can live -- counts
them, and bounds the band they live in (3 |sum| + 2 ulp of u).  ``ptswap.npz`` holds ~10^4 generic decisions and no tie; this is the missing case."""
import numpy as np

from oracle import oracle as orc

N = 10_000_000


def _log_n(u):
    out = np.empty_like(u)
    orc.lib().orc_log_n(u.size, orc._p(u), orc._p(out))
    return out


def _band(la):
    """Half-width, in ulp of u, of the band around u = exp(sum) inside which the two forms may differ: log u is compared at the
    resolution of `sum` (one ulp of the sum is |sum| 2^-52 relative in u, i.e. up to 2 |sum| ulp of u), each of log and exp is
    good to an ulp of its own result, and the sum's own half-ulp is another |sum| ulp of u: 3 |sum| + 2 in all."""
    return 3.0 * np.abs(la) + 2.0


def test_log_space_swap_test_agrees_with_the_reference_form_outside_the_rounding_band():
    rs = np.random.RandomState(20260930)
    # sums of a swap test that is not decided by its sign: u <= exp(sum) can only fail for sum < 0; a ladder step of
    # 1 + sqrt(2 / d) at d = 100 gives |sum| of order 1, the hot end reaches tens
    la = -np.concatenate([rs.uniform(0.0, 2.0, N // 2), 10.0 ** rs.uniform(-6.0, 1.6, N - N // 2)])
    e = np.exp(la)                                              # the reference's threshold (libm)
    # u = exp(sum) displaced by up to twice the band, either way: the adversarial near-ties
    half = np.ceil(2.0 * _band(la)).astype(np.int64)
    k = (rs.randint(0, 1 << 30, N) % (2 * half + 1)) - half
    u = (e.view(np.int64) + k).view(np.float64)
    ok = (u > 0.0) & (u < 1.0)                                  # uniforms of [0, 1); u = 0 accepts in both forms
    u, la, e, k = u[ok], la[ok], e[ok], k[ok]
    ref = u <= e                                                # PTMCMCSampler.py:679
    mine = _log_n(np.ascontiguousarray(u)) <= la                # oracle / HIP sweep
    differ = ref != mine
    inside = np.abs(k) <= _band(la)
    n, nd = u.size, int(differ.sum())
    # (1) outside the band the two forms ALWAYS agree
    assert not differ[~inside].any(), "a disagreement %d ulp from the tie at sum = %g" % (
        np.abs(k[differ & ~inside]).max(), la[differ & ~inside][0])
    # (2) inside it they differ for a minority of the near-ties (the two roundings land on opposite sides of the tie)
    cond = nd / max(1, int(inside.sum()))
    assert cond < 0.5, cond
    # (3) what that means for a run: a 53-bit uniform falls into the band around exp(sum) with probability
    # (2 band + 1) 2^-53, so the chance that ONE swap decision differs from the reference's is below that times the conditional
    # rate measured here -- of order 1e-15 at |sum| ~ 1 (one decision in 1e15: at the 2.6e8 pair tests per second of the
    # 64 x 4096 benchmark, once in a month and a half of sampling), 1e-14 at the hot end of a long ladder
    typical = (2 * _band(-1.0) + 1) * 2.0 ** -53 * cond
    hot = (2 * _band(-40.0) + 1) * 2.0 ** -53 * cond
    print("near-ties: %d, inside the band: %d, disagreements: %d (%.3f of those inside); per swap decision of a real run: "
          "< %.1e at |sum| = 1, < %.1e at |sum| = 40" % (n, int(inside.sum()), nd, cond, typical, hot))
    assert typical < 2e-15 and hot < 3e-14
    # (4) the extremes agree exactly: u = 0 accepts in both forms, a NaN sum accepts in neither
    assert (0.0 <= np.exp(-5.0)) and (orc.lib().orc_log(0.0) <= -5.0)
    assert not (0.3 <= np.exp(np.nan)) and not (orc.lib().orc_log(0.3) <= np.nan)


def test_sweep_decisions_with_explicit_uniforms_follow_the_reference_form():
    """orc_swap_sweep replaying uniforms placed 16+ ulp on either side of every pair's threshold takes exactly the decisions of
    the reference's ``u <= exp(sum)`` loop (PTMCMCSampler.py:666-686), restated here in NumPy."""
    rs = np.random.RandomState(7)
    for n in (2, 5, 64, 512):
        ladder = (1 + np.sqrt(2.0 / 100)) ** np.arange(n)
        lnL = -50.0 * ladder * (1 + 0.1 * rs.randn(n))
        # reference sweep, choosing every uniform relative to the pair's own threshold as it goes
        m = np.arange(n)
        us, want_acc = [], np.zeros(n)
        for k in range(n - 2, -1, -1):
            la = -lnL[m[k]] / ladder[k]
            la += -lnL[m[k + 1]] / ladder[k + 1]
            la += lnL[m[k + 1]] / ladder[k]
            la += lnL[m[k]] / ladder[k + 1]
            thr = np.exp(la)
            side = rs.randint(2)
            if thr >= 1.0:
                u = rs.rand()
            else:
                u = float((np.float64(thr).view(np.int64) + (16 if side else -16)).view(np.float64))
                u = min(max(u, 0.0), 1 - 2.0 ** -53)
            us.append(u)
            if u <= thr:
                m[[k, k + 1]] = m[[k + 1, k]]
                want_acc[k] += 1
        mo, acc = orc.swap_sweep(ladder, lnL, uniforms=np.asarray(us))
        assert np.array_equal(mo[0], m) and np.array_equal(acc[0].astype(float), want_acc), n
