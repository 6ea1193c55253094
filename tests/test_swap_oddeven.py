"""Odd/even swap mode (include/ptmi.h PTMI_SWAP_ODDEVEN; an engine mode, not in the reference).

CPU: the oracle's odd/even function equals the reference-pinned sweep (tests/test_oracle_golden.py pins that one to
PTswap) with the untried pairs' uniforms forced to "never accept"; pairs alternate with the swap epoch.
GPU: the parallel kernel, the masked sweep of a sharded ladder and the facade run bit-exact against the oracle."""
import numpy as np
import pytest

from oracle import oracle as orc

SLOT_SWAP = 0x10000


def _pair_uniform(seed, it, walker, n, k):
    w = orc.philox([it & 0xFFFFFFFF, it >> 32, walker * n, SLOT_SWAP + k], [seed & 0xFFFFFFFF, seed >> 32])
    return ((((w[1] << 32) | w[0]) >> 11) * 2.0 ** -53)


@pytest.mark.parametrize("n", [2, 3, 8, 65])
def test_oddeven_is_the_sweep_with_untried_pairs_suppressed(n):
    rng = np.random.default_rng(n)
    W, seed, it = 9, 0x1234567890, 700
    ladder = orc.temperature_ladder(n, 10)
    L = rng.normal(size=(W, n)) * 3.0
    for parity in (0, 1):
        m, acc = orc.swap_oddeven(ladder, L, parity, it=it, seed=seed, walker0=2)
        tried = 0
        for w in range(W):
            # the sweep draws hottest pair first: k = n-2 .. 0
            us = [_pair_uniform(seed, it, 2 + w, n, k) if (k & 1) == parity else np.inf for k in range(n - 2, -1, -1)]
            ms, accs = orc.swap_sweep(ladder, L[w], uniforms=us)
            assert np.array_equal(ms[0], m[w]) and np.array_equal(accs[0], acc[w])
            assert all(acc[w, k] == 0 for k in range(n) if (k & 1) != parity)
            tried += sum(1 for k in range(n - 1) if (k & 1) == parity)
        assert tried == W * ((n - parity) // 2)
        assert np.array_equal(np.sort(m, axis=1), np.tile(np.arange(n), (W, 1)))
    if n > 3:
        assert acc.sum() > 0


def test_parity_alternates_with_the_swap_epoch():
    assert [orc.swap_parity(it, 10) for it in (10, 20, 30, 40)] == [1, 0, 1, 0]
    o = orc.OracleEngine(4, 6, 3, np.eye(4) * 0.1, weights=(20, 0, 0), cov_update=50, burn=100, tskip=5, seed=3,
                         swap_mode="oddeven")
    o.init_state(np.random.default_rng(0).normal(size=(3, 6, 4)))
    o.run(100)
    assert o.swap_proposed == 20
    assert (o.nswap[:, :5] > 0).all() and (o.nswap[:, 5] == 0).all()     # every pair gets its turns
    assert np.array_equal(np.sort(o.slot_of, axis=1), np.tile(np.arange(6), (3, 1)))


@pytest.mark.gpu
@pytest.mark.parametrize("d,nt,W,cov_mode", [(5, 4, 6, "per_walker"), (100, 7, 5, "pooled"), (20, 64, 9, "per_walker")])
def test_gpu_oddeven_bit_exact(d, nt, W, cov_mode):
    from ptmcmcsampler_amd.engine import PTEngine
    kw = dict(weights=(20, 20, 20), cov_update=50, burn=100, tskip=10, seed=21, cov_mode=cov_mode, swap_mode="oddeven")
    cov0 = np.eye(d) * 0.05
    p0 = np.random.default_rng(d).normal(size=(W, nt, d)) * 0.4
    g, o = PTEngine(d, nt, W, cov0, **kw), orc.OracleEngine(d, nt, W, cov0, **kw)
    g.init_state(p0)
    o.init_state(p0)
    g.run(330)
    o.run(330)
    g.sync()
    for name in ("X", "lnL", "lp", "temp_of", "slot_of", "nacc", "jstat", "nswap", "AM", "M2", "Ut"):
        assert np.array_equal(np.asarray(g.get(name)).view(np.uint64) if name in ("X", "lnL", "lp", "AM", "M2", "Ut") else g.get(name),
                              np.asarray(getattr(o, name)).view(np.uint64) if name in ("X", "lnL", "lp", "AM", "M2", "Ut") else getattr(o, name)), name
    assert g.swap_proposed == o.swap_proposed == 33 and o.nswap.sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 4])
def test_gpu_oddeven_sharded_emulated_ranks(nranks):
    """Sharded ladder in odd/even mode (masked sweep on every block + device exchange) == single engine == oracle,
    and an epoch moves rows across a block edge only when the edge pair is tried."""
    import os
    import sys
    import threading
    sys.path.insert(0, os.path.dirname(__file__))
    from thread_comm import ThreadComm, ThreadWorld
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    d, ntb, W, n = 10, 3, 17, 230
    ntg = ntb * nranks
    kw = dict(weights=(20, 20, 20), cov_update=50, burn=100, tskip=10, seed=5, cov_mode="per_walker", swap_mode="oddeven")
    cov0 = np.eye(d) * 0.05
    p0 = np.random.RandomState(3).randn(W, ntg, d) * 0.4
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
        assert L.exchange_violations() == 0
        so = L.get("slot_of")
        bt = lambda a: np.take_along_axis(a, so.reshape(so.shape + (1,) * (a.ndim - 2)), axis=1)
        assert np.array_equal(bt(L.get("X")), ref.by_temp(ref.X)[:, sl])
        assert np.array_equal(bt(L.get("lnL")), ref.by_temp(ref.lnL)[:, sl])
        assert np.array_equal(L.get("nswap")[:, sl], ref.nswap[:, sl])
    assert ref.nswap[:, ntb - 1].sum() > 0, "the edge pair never accepted: the test would prove nothing"
