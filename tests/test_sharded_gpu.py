"""The sharded engine on a real GPU over RCCL (world_size 1: the only size a 1-GPU box allows).
Exercises the three-piece swap kernels and the all-gather / all-to-all plumbing on device tensors;
the multi-rank exchange logic itself is proven on CPU in test_sharded_gloo.py."""
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def world1():
    import torch
    import torch.distributed as dist
    from ptmcmcsampler_amd import _lib
    _lib.load()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("cov_mode", ["per_walker", "pooled"])
def test_sharded_world1_equals_single_engine_and_oracle(world1, cov_mode):
    from oracle import oracle as orc
    from ptmcmcsampler_amd.engine import PTEngine
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    d, nt, W, n = 12, 6, 9, 330
    kw = dict(weights=(20, 20, 20), cov_update=50, burn=100, tskip=10, seed=31, cov_mode=cov_mode)
    cov0 = np.eye(d) * 0.05
    p0 = np.random.RandomState(2).randn(W, nt, d) * 0.4
    s = ShardedPTEngine(d, nt, W, cov0, group=world1.group.WORLD, **kw)
    g = PTEngine(d, nt, W, cov0, **kw)
    o = orc.OracleEngine(d, nt, W, cov0, **kw)
    for e in (s, g, o):
        e.init_state(p0)
        e.run(n)
    s.sync()
    g.sync()
    for name in ("X", "lnL", "lp", "temp_of", "slot_of", "nacc", "jstat", "nswap", "AM", "Ut", "S"):
        a, b, c = s.get(name), g.get(name), getattr(o, name)
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), name
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(c).view(np.uint8)), name
    assert s.swap_proposed == g.swap_proposed == n // 10


@pytest.mark.parametrize("nranks,cov_mode,ntb", [(2, "per_walker", 3), (4, "pooled", 3), (8, "per_walker", 3), (2, "pooled", 70), (4, "pooled", 70)])
def test_device_exchange_with_emulated_ranks(nranks, cov_mode, ntb):
    """Several temperature blocks on ONE GPU (one thread per rank, in-process communicator): the device-side
    exchange (ptmi_swap_sweep_blocks / ptmi_exchange_pack / ptmi_exchange_apply) must reproduce the single-engine
    run and the oracle bit for bit, with rows really crossing block edges."""
    import sys
    import threading
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from thread_comm import ThreadComm, ThreadWorld
    from oracle import oracle as orc
    from ptmcmcsampler_amd.engine import PTEngine
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    d, W, n = (10, 33, 330) if ntb < 64 else (4, 7, 130)       # ntb > 64: more local ranks than lanes in the plan kernel
    ntg = ntb * nranks
    kw = dict(weights=(20, 20, 20), cov_update=50, burn=100, tskip=10, seed=5, cov_mode=cov_mode)
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
            assert e.device_exchange
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
    moved = 0
    for r, e in enumerate(engines):
        L, sl = e.local, slice(r * ntb, (r + 1) * ntb)
        assert L.exchange_violations() == 0
        for name in ("X", "lnL", "lp"):
            assert np.array_equal(L.by_temp(name), ref.by_temp(getattr(ref, name))[:, sl]), (r, name)
        assert np.array_equal(L.get("nacc"), ref.nacc[:, sl]) and np.array_equal(L.get("jstat"), ref.jstat[:, sl])
        assert np.array_equal(L.get("nswap")[:, sl], ref.nswap[:, sl])
        assert np.array_equal(L.get("Ut"), ref.Ut) and np.array_equal(L.get("S"), ref.S)
        if r == 0:
            assert np.array_equal(L.get("AM"), ref.AM) and np.array_equal(L.get("M2"), ref.M2)
        assert e.swap_proposed == n // 10
        # rows that are not where they started prove that states crossed block edges
        moved += int((np.abs(L.by_temp("X") - p0[:, sl]).sum(-1) > 0).sum())
    assert ref.nswap[:, ntb - 1].sum() > 0, "no swap was ever accepted across the first block edge"
    # transport: every rank took the same decision at every swap epoch; the neighbour links alone served some epochs, and
    # with three-rank blocks some sweep carried a row across a whole block (the all-to-all fallback)
    hops = {e.neighbour_swaps for e in engines}
    assert len(hops) == 1 and 0 <= engines[0].neighbour_swaps <= n // 10
    if nranks == 2:
        assert engines[0].neighbour_swaps == n // 10          # two blocks have no one but each other
    # (4 x 70 ranks at d = 4: the ladder's hot end is flat, every pair there accepts and the carried state crosses whole
    # blocks -- all epochs take the all-to-all; the case is here for the 280-rank sweep, 32 walkers per block)
    if ntb == 3 and nranks >= 4:
        assert engines[0].neighbour_swaps < n // 10           # 33 walkers on three-rank blocks: some sweep always crosses one


def _run_emulated_ladder(nranks, ntb, d, W, n, cov0, p0, kw):
    """One thread per temperature block on the one GPU (thread_comm.ThreadComm) + the whole ladder on the oracle."""
    import sys
    import threading
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from thread_comm import ThreadComm, ThreadWorld
    from oracle import oracle as orc
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    ntg = ntb * nranks
    ref = orc.OracleEngine(d, ntg, W, cov0, **kw)
    ref.init_state(p0)
    ref.run(n)
    world = ThreadWorld(nranks)
    engines, errs = [None] * nranks, []

    def rank_main(r):
        try:
            e = ShardedPTEngine(d, ntg, W, cov0, comm=ThreadComm(world, r), **kw)
            assert e.device_exchange
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
    return ref, engines


def _check_blocks(ref, engines, ntb, n_swaps, gj=False):
    for r, e in enumerate(engines):
        L, sl = e.local, slice(r * ntb, (r + 1) * ntb)
        assert L.exchange_violations() == 0
        for name in ("X", "lnL", "lp"):
            assert np.array_equal(L.by_temp(name), ref.by_temp(getattr(ref, name))[:, sl]), (r, name)
        assert np.array_equal(L.get("nacc"), ref.nacc[:, sl]) and np.array_equal(L.get("jstat"), ref.jstat[:, sl]), r
        assert np.array_equal(L.get("nswap")[:, sl], ref.nswap[:, sl]), r
        assert np.array_equal(L.get("Ut"), ref.Ut) and np.array_equal(L.get("S"), ref.S), r
        if gj:
            assert np.array_equal(L.get("gj"), ref.gj[:, sl]), r                   # a rank's jump-object state
        if r == 0:
            assert np.array_equal(L.get("AM"), ref.AM) and np.array_equal(L.get("M2"), ref.M2)
        assert e.swap_proposed == n_swaps
    edges = ref.nswap[:, ntb - 1::ntb][:, :len(engines) - 1].sum(0)
    assert (edges > 0).all(), "a block edge no swap ever crossed: %r" % (edges,)
    assert len({e.neighbour_swaps for e in engines}) == 1                    # every rank chose the same transport at every epoch


@pytest.mark.parametrize("start", ["equilibrium", "flat"])
def test_config4_ladder_8_blocks_of_64_ranks_1000d(start):
    """BASELINE configs[3] as it is sharded: 1000-d isotropic Gaussian, 512 temperature ranks = 8 blocks of 64 (one per
    GPU; here eight emulated ranks on the one GPU), default SCAM/AM/DE cycle, pooled covariance, against the oracle's
    single ladder, bit for bit: 8 KB rows packed / applied across all seven block edges (PTMCMCSampler.py:631-697),
    the 8 MB factorization and the new DE rows broadcast from block 0 (:545-576), 64 lanes per chain.
    ``equilibrium``: every rank starts from a draw of its own tempered target -- pairs accept at the ladder's design rate,
    no state crosses a whole block, all epochs go over the two neighbour links.  ``flat``: all ranks start near the mode,
    every pair accepts and the hottest state travels to rank 0 in one sweep -- the multi-hop epochs (all-to-all)."""
    from ptmcmcsampler_amd import _lib
    from ptmcmcsampler_amd.ladder import temperature_ladder
    d, nranks, ntb, W, n = 1000, 8, 64, 8, 50
    ntg = nranks * ntb
    kw = dict(weights=(20, 4, 20), cov_update=20, burn=40, tskip=10, seed=5, cov_mode="pooled")
    cov0 = np.eye(d) * 0.01
    rs = np.random.RandomState(3)
    if start == "equilibrium":
        p0 = rs.randn(W, ntg, d) * np.sqrt(temperature_ladder(ntg, d))[None, :, None]
    else:
        p0 = rs.randn(W, ntg, d) * 0.1
    ref, engines = _run_emulated_ladder(nranks, ntb, d, W, n, cov0, p0, kw)
    _check_blocks(ref, engines, ntb, n // 10)
    flags, G, E = engines[3].local.last_variant()
    assert G == 64 and flags & _lib.VAR_FULL
    assert ref.jstat[..., 2, 0].sum() > 0 and ref.jstat[..., 1, 1].sum() > 0        # DE became active; AM jumps were accepted
    if start == "equilibrium":
        assert engines[0].neighbour_swaps == n // 10
    else:
        assert engines[0].neighbour_swaps < n // 10


def test_config5_ladder_8_blocks_of_16_ranks_curved_nuts():
    """BASELINE configs[4] as it is sharded: 20-d curved likelihood, 128 ranks = 8 blocks of 16, SCAM + DE + NUTS cycle with
    the box prior, against the oracle's single ladder, bit for bit, incl. every rank's NUTS step-size state (the jump
    objects belong to ranks, nutsjump.py:379-433, and stay put while states cross the block edges)."""
    from ptmcmcsampler_amd import _lib
    d, nranks, ntb, W, n = 20, 8, 16, 9, 120
    ntg = nranks * ntb
    kw = dict(logl=("curved",), logp=("box", np.full(d, -10.0), np.full(d, 10.0)), weights=(10, 0, 10), grad_weights=(10, 0),
              cov_update=30, burn=60, tskip=10, seed=7, cov_mode="pooled")
    cov0 = np.eye(d)
    rs = np.random.RandomState(5)
    p0 = np.array([-0.1, -0.5] * (d // 2)) + rs.randn(W, ntg, d) * 0.3 * np.linspace(0.2, 3.0, ntg)[None, :, None]
    p0 = np.clip(p0, -9.5, 9.5)
    ref, engines = _run_emulated_ladder(nranks, ntb, d, W, n, cov0, p0, kw)
    _check_blocks(ref, engines, ntb, n // 10, gj=True)
    flags, G, E = engines[5].local.last_variant()
    assert flags & _lib.VAR_GRADJUMP
    assert ref.jstat[..., 3, 0].sum() > 0 and ref.jstat[..., 2, 0].sum() > 0        # NUTS and DE proposals were made


@pytest.mark.parametrize("eig_mode,lag,stats_async", [("sytrd", 2, False), ("sytrd", 3, True), ("hipsolver", 2, False), ("lapack", 2, True),
                                                       ("sytrd", 3, False), ("sytrd", 4, False),        # a period is three launches: the late finish
                                                       ("ql", 2, False), ("ql", 4, False)])             # ptmi_eig_ql_from on the owner's side stream (ndim <= 128)
def test_sharded_ladder_with_the_owner_factorizing_on_its_side_stream(eig_mode, lag, stats_async):
    """eig_lag = L with the factorization on the owner's side (PTMCMCSampler.py:545-560; ShardedPTEngine): the block that holds rank 0
    runs statistics + ptmi_eig_sytrd / the library's eigensolver (with stats_async the statistics too on the side stream, two AM
    rings) while EVERY block runs L more launches with the table in force; the new table is broadcast behind the L-th launch's swap.
    Four emulated ranks on the one GPU against the single-engine run with the same eig_lag: bit for bit -- and the single-engine run
    against OracleEngine(eig_lag=L): the host's and the QL factorization bit for bit, the other device factorizations through the
    tables they made (each decomposes the oracle's covariance to 1e-12, the chains stepped with it equal the device's)."""
    import sys
    import threading
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from thread_comm import ThreadComm, ThreadWorld
    from oracle import oracle as orc
    from ptmcmcsampler_amd.engine import PTEngine
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    d, nranks, ntb, W, n = (60 if eig_mode == "ql" else 300), 4, 4, 6, 170
    ntg = nranks * ntb
    kw = dict(weights=(20, 0, 20), cov_update=30, burn=60, tskip=10, seed=11, cov_mode="pooled", eig_mode=eig_mode, eig_lag=lag)
    cov0 = np.eye(d) * 0.01
    p0 = np.random.RandomState(3).randn(W, ntg, d) * 0.2
    if eig_mode == "lapack":
        ref = orc.OracleEngine(d, ntg, W, cov0, **kw)
    else:
        ref = PTEngine(d, ntg, W, cov0, **kw)
    ref.init_state(p0)
    if eig_mode == "lapack":
        ref.run(n)
    else:
        # the single engine the blocks are compared with is itself held to the oracle: with the restated device solver ("ql") bit for
        # bit, tables included; with the solvers the oracle does not restate ("sytrd", "hipsolver") stepped on the tables the device
        # made, each checked against the oracle's covariance (test_device_tables_gpu.lockstep)
        from test_device_tables_gpu import TableTap, lockstep
        from test_gpu_parity import _compare, assert_same
        okw = dict(kw, eig_mode="ql" if eig_mode == "ql" else "lapack")
        o = orc.OracleEngine(d, ntg, W, cov0, **okw)
        o.init_state(p0)
        if eig_mode == "ql":
            ref.run(n)
            o.run(n)
            assert_same(ref.get("Ut"), o.Ut, "single engine vs oracle: Ut")
        else:
            assert lockstep(ref, o, TableTap(ref), n, "single engine, %s lag %d" % (eig_mode, lag)) >= 4
            ref.mh_steps = type(ref).mh_steps.__get__(ref)
        _compare(ref, o, "single engine vs oracle (%s, lag %d): " % (eig_mode, lag))
        assert_same(ref.get("cov"), o.cov, "single engine vs oracle: cov")
        ref.sync()
    rget = (lambda name: getattr(ref, name)) if eig_mode == "lapack" else ref.get
    rby = (lambda name: ref.by_temp(getattr(ref, name))) if eig_mode == "lapack" else ref.by_temp
    world = ThreadWorld(nranks)
    engines, errs = [None] * nranks, []

    def rank_main(r):
        try:
            e = ShardedPTEngine(d, ntg, W, cov0, comm=ThreadComm(world, r), stats_async=stats_async, **kw)
            assert e.eig_lag == lag and e.local.stats_async == (stats_async and r == 0)
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
        for name in ("X", "lnL", "lp"):
            assert np.array_equal(L.by_temp(name), rby(name)[:, sl]), (r, name)
        assert np.array_equal(L.get("nacc").astype(np.int64), np.asarray(rget("nacc")).astype(np.int64)[:, sl]), r
        assert np.array_equal(L.get("Ut"), rget("Ut")) and np.array_equal(L.get("S"), rget("S")), r
        assert np.array_equal(np.roll(L.get("DE")[0], -L.de_head, axis=0),
                              rget("DE")[0] if eig_mode == "lapack" else np.roll(ref.get("DE")[0], -ref.de_head, axis=0)), r
    assert engines[0].local.eig_epochs >= 4


def test_two_real_processes_share_the_gpu_over_gloo():
    """True multi-process run of ShardedPTEngine + DistComm (two ranks on cuda:0, gloo moving device tensors),
    both swap modes, against the oracle: tools/two_proc_one_gpu.py."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "two_proc_one_gpu.py"), "2"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "all ranks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("extra,ntemps,ngpus", [((), 16, 2), (("--ndim", "1000", "--nwalkers", "64", "--mix", "default"), 64, 2),
                                                ((), 16, 4), (("--ndim", "1000", "--nwalkers", "32"), 64, 3)])
def test_bench_launches_its_own_ranks(extra, ntemps, ngpus):
    """`python bench.py --gpus 2` as the driver calls it (second case: BASELINE configs[3]'s shape, 1000-d, 64 ranks per GPU): the script starts its two ranks itself.  A one-GPU box cannot give
    RCCL two devices, so the rehearsal runs over gloo with both ranks on cuda:0 (PTMI_DIST_BACKEND=gloo); the sharded
    engine, the neighbour send/recv at the block edge and the JSON contract are the ones of the RCCL run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PTMI_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--nwalkers", "128", "--ntemps", "16"] if not extra else list(extra)
    # (three and four ranks: the lnL gather as one grouped send/recv to every peer, DistComm.all_gather; the 1000-d SCAM case: the
    # owner's device factorization with eig_lag = a covariance period, the table broadcast behind the next epoch's statistics)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ngpus), "--steps", "12", "--warmup", "2", "--no-cpu-baseline",
                        "--ess-burn", "0", "--ess-window", "3000"] + args,
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ngpus and out["steps"] == 12 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["rccl_ranks"] == 0                       # gloo rehearsal: no RCCL ranks
    assert out["config"]["parallelism"] == "temperature blocks x%d" % ngpus and out["config"]["ntemps_per_gpu"] == ntemps
    if extra:
        assert out["config"]["ndim"] == 1000 and out["config"]["nwalkers"] == int(extra[3])
    assert out["swap_epochs_timed"] == 12 and out["cov_epochs_timed"] == 1 and out["swap_accept_rate_pair0"] > 0
    # the pre-flight: a small ladder sharded over THESE ranks against one engine, every rank's block bit for bit, rows across every edge
    det = out["sharded_selfcheck_detail"]
    assert out["sharded_selfcheck"] is True and det["blocks_equal_single_engine"] == [True] * ngpus
    assert det["edge_swaps_accepted_min_over_edges"] > 0 and det["swap_epochs"] == 30 and 0 <= det["neighbour_swaps"] <= 30
    assert out["roofline"]["frac"] <= 1.0
    # the ESS leg ran (after the timed region) and FLAGS its 3000-iteration window as too short to trust
    assert out["ess_per_sec"] > 0 and out["tau_int"] >= 1 and out["ess_window_iters"] == 3000 and out["ess_window_ok"] is False
