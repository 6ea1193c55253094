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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as the driver calls it: the script starts its two ranks itself.  A one-GPU box cannot give
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
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2", "--nwalkers", "128",
                        "--ntemps", "16", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["rccl_ranks"] == 0                       # gloo rehearsal: no RCCL ranks
    assert out["config"]["parallelism"] == "temperature blocks x2" and out["config"]["ntemps_per_gpu"] == 16
    assert out["swap_epochs_timed"] == 12 and out["cov_epochs_timed"] == 1 and out["swap_accept_rate_pair0"] > 0
    assert out["ess_per_sec"] is not None and out["roofline"]["frac"] <= 1.0
