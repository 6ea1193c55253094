"""Two (or more) real processes sharing cuda:0, torch.distributed over gloo with device tensors: exercises
ShardedPTEngine + DistComm under true multi-process semantics on a one-GPU box (RCCL refuses two ranks on one
device).  Usage: python tools/two_proc_one_gpu.py [world]"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from oracle import oracle as orc
    from ptmcmcsampler_amd.sharded import ShardedPTEngine
    d, ntb, W, n = 10, 3, 17, 230
    ntg = ntb * world
    for swap_mode in ("sweep", "oddeven"):
        kw = dict(weights=(20, 20, 20), cov_update=50, burn=100, tskip=10, seed=5, cov_mode="per_walker", swap_mode=swap_mode)
        cov0 = np.eye(d) * 0.05
        p0 = np.random.RandomState(3).randn(W, ntg, d) * 0.4
        e = ShardedPTEngine(d, ntg, W, cov0, group=dist.group.WORLD, **kw)
        e.init_state(p0)
        e.run(n)
        e.sync()
        ref = orc.OracleEngine(d, ntg, W, cov0, **kw)
        ref.init_state(p0)
        ref.run(n)
        L, sl = e.local, slice(rank * ntb, (rank + 1) * ntb)
        so = L.get("slot_of")
        bt = lambda a: np.take_along_axis(a, so.reshape(so.shape + (1,) * (a.ndim - 2)), axis=1)
        assert np.array_equal(bt(L.get("X")), ref.by_temp(ref.X)[:, sl]), "X"
        assert np.array_equal(bt(L.get("lnL")), ref.by_temp(ref.lnL)[:, sl]), "lnL"
        assert np.array_equal(L.get("nswap")[:, sl], ref.nswap[:, sl]), "nswap"
        print("rank %d/%d %s ok (device exchange: %s, rows moved %d)" % (rank, world, swap_mode, e.device_exchange, e.rows_moved), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)
    print("all ranks ok")
