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
