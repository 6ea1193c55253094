"""Temperature-block sharding over world_size 2 and 4 (gloo, CPU): the boundary exchange of
ptmcmcsampler_amd/sharded.py must reproduce the single-process run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cov_mode, swap_mode, out_dir, eig_lag=0, eig_mode="lapack"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from oracle_local import OracleLocal
        from ptmcmcsampler_amd.sharded import ShardedPTEngine
        d, ntg, W, n = 6, 8, 5, 330
        kw = dict(weights=(20, 20, 20), cov_update=50, burn=100, tskip=10, seed=99, cov_mode=cov_mode, swap_mode=swap_mode, eig_lag=eig_lag,
                  eig_mode=eig_mode)
        rs = np.random.RandomState(1)
        cov0 = np.eye(d) * 0.05
        p0 = rs.randn(W, ntg, d) * 0.5
        e = ShardedPTEngine(d, ntg, W, cov0, group=dist.group.WORLD, local_factory=OracleLocal, **kw)
        e.init_state(p0)
        e.run(n)
        ref = orc.OracleEngine(d, ntg, W, cov0, **kw)
        ref.init_state(p0)
        ref.run(n)
        nt, t0 = e.nt, e.temp0
        o = e.local.o
        sl = slice(t0, t0 + nt)
        assert np.array_equal(o.by_temp(o.X), ref.by_temp(ref.X)[:, sl]), "states by temperature"
        assert np.array_equal(o.by_temp(o.lnL), ref.by_temp(ref.lnL)[:, sl])
        assert np.array_equal(o.by_temp(o.lp), ref.by_temp(ref.lp)[:, sl])
        assert np.array_equal(o.nacc, ref.nacc[:, sl]) and np.array_equal(o.jstat, ref.jstat[:, sl])
        assert np.array_equal(o.nswap[:, sl], ref.nswap[:, sl])
        assert np.array_equal(o.Ut, ref.Ut) and np.array_equal(o.S, ref.S)
        assert np.array_equal(np.roll(e.local.ring, -e.local.head, axis=1), ref.DE)
        if rank == 0:
            assert np.array_equal(o.AM, ref.AM) and np.array_equal(o.M2, ref.M2)
        assert e.swap_proposed == ref.swap_proposed == n // 10
        moved = torch.tensor([e.rows_moved])
        dist.all_reduce(moved)
        assert int(moved) > 0, "no row ever crossed a block edge: the test would prove nothing"
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("%d" % int(moved))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cov_mode,swap_mode", [(2, "per_walker", "sweep"), (2, "pooled", "sweep"), (4, "per_walker", "sweep"),
                                                      (2, "per_walker", "oddeven")])
def test_sharded_equals_single_process(tmp_path, world, cov_mode, swap_mode):
    mp.spawn(_worker, args=(world, _free_port(), cov_mode, swap_mode, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok_%d" % r for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_with_the_factorization_one_launch_late(tmp_path, world):
    """eig_lag = 1 on a sharded ladder: the owner of rank 0 factorizes the pooled covariance while every block runs the launch that
    follows the epoch, the table is broadcast behind that launch's swap -- bit for bit the single-process OracleEngine(eig_lag=1)."""
    mp.spawn(_worker, args=(world, _free_port(), "pooled", "sweep", str(tmp_path), 1), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok_%d" % r for r in range(world)]


@pytest.mark.parametrize("world,lag", [(2, 3), (4, 2), (4, 7)])
def test_sharded_with_a_device_factorization_several_launches_late(tmp_path, world, lag):
    """eig_lag = L with a factorization that is not the host's (here the restated device QL solver, orc_eig_ql; on the GPUs
    ptmi_eig_sytrd / the library on the owner's side stream): every block runs L more launches with the table in force, the owner's
    new table is broadcast behind the L-th launch's swap (L = 7 > the five launches of a covariance period: the next epoch finishes
    it first) -- bit for bit the single-process OracleEngine(eig_lag=L, eig_mode="ql")."""
    mp.spawn(_worker, args=(world, _free_port(), "pooled", "sweep", str(tmp_path), lag, "ql"), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok_%d" % r for r in range(world)]


def test_plan_exchange_is_a_permutation():
    """Property test of the exchange plan on random permutations (no process group needed)."""
    sys.path.insert(0, ROOT)
    from ptmcmcsampler_amd.sharded import plan_exchange
    rs = np.random.RandomState(0)
    W, world, nt = 7, 4, 3
    ntg = world * nt
    m = np.stack([rs.permutation(ntg) for _ in range(W)]).astype(np.int32)
    slot = [np.stack([rs.permutation(nt) for _ in range(W)]).astype(np.int32) for _ in range(world)]
    plans = [plan_exchange(torch.from_numpy(m), torch.from_numpy(slot[r]), r * nt, nt, r, world) for r in range(world)]
    for r in range(world):
        ns = plans[r]["new_slot"].numpy()
        assert (np.sort(ns, axis=1) == np.arange(nt)).all()
        for q in range(world):
            assert plans[r]["send_counts"][q] == plans[q]["recv_counts"][r]


def _p2p_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ptmcmcsampler_amd.sharded import DistComm
        comm = DistComm(dist.group.WORLD)
        W, k = 5, 4
        send = torch.full((world, W, k), -1.0, dtype=torch.float64)
        for q in range(world):
            send[q] = 100.0 * rank + q + torch.arange(W * k, dtype=torch.float64).reshape(W, k) / 1000.0
        recv = torch.full((world, W, k), -7.0, dtype=torch.float64)
        comm.neighbour_exchange(send, recv)
        for q in range(world):
            if abs(q - rank) == 1:
                want = 100.0 * q + rank + torch.arange(W * k, dtype=torch.float64).reshape(W, k) / 1000.0
                assert torch.equal(recv[q], want), (rank, q)
            else:
                assert bool((recv[q] == -7.0).all()), (rank, q)          # nothing arrives from anyone but the neighbours
        open(os.path.join(out_dir, "p2p_%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_neighbour_exchange_moves_only_the_two_edge_segments(tmp_path, world):
    """DistComm.neighbour_exchange = the grouped send/recv of the block-edge rows (what runs as ncclSend/ncclRecv on RCCL)."""
    mp.spawn(_p2p_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["p2p_%d" % r for r in range(world)]
