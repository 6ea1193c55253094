"""In-process communicator: N threads of one process play N ranks (tests only).

Lets ONE GPU run the whole temperature-block exchange of ShardedPTEngine -- HIP kernels included -- with several
blocks, which a 1-GPU box cannot do over RCCL (one rank per device)."""
import threading

import torch


class ThreadWorld(object):
    def __init__(self, n):
        self.n = n
        self.bar = threading.Barrier(n)
        self.slots = [None] * n


class ThreadComm(object):
    def __init__(self, world, rank):
        self.w, self.rank, self.world = world, rank, world.n

    def _sync(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.w.bar.wait()

    def all_gather(self, out, inp):
        self.w.slots[self.rank] = inp
        self._sync()
        out.copy_(torch.cat([s.reshape(-1, inp.shape[-1]) for s in self.w.slots]).reshape(out.shape))
        self._sync()

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        assert out_splits is None and in_splits is None, "equal splits only"
        self.w.slots[self.rank] = inp
        self._sync()
        for q in range(self.world):
            out[q].copy_(self.w.slots[q][self.rank])
        self._sync()

    def broadcast(self, t):
        if self.rank == 0:
            self.w.slots[0] = t
        self._sync()
        if self.rank != 0:
            t.copy_(self.w.slots[0])
        self._sync()

    def neighbour_exchange(self, send, recv):
        """send[q] -> rank q, recv[q] <- rank q for the two neighbouring ranks only."""
        self.w.slots[self.rank] = send
        self._sync()
        for q in (self.rank - 1, self.rank + 1):
            if 0 <= q < self.world:
                recv[q].copy_(self.w.slots[q][self.rank])
        self._sync()
