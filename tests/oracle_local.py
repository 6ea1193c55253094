"""CPU stand-in for the per-GPU engine, backed by the oracle (tests only).

Gives ``ShardedPTEngine`` the same interface as ``PTEngine`` on CPU torch tensors that
alias the oracle's numpy arrays, so the block-boundary exchange logic can be exercised with
world_size > 1 over gloo without a GPU."""
import ctypes as C

import numpy as np
import torch

from oracle import oracle as orc


class OracleLocal(object):
    def __init__(self, ndim, ntemps, nwalkers, cov0, **kw):
        kw.pop("device", None)
        self.o = o = orc.OracleEngine(ndim, ntemps, nwalkers, cov0, **kw)
        self.device = torch.device("cpu")
        self.owns_cold = o.temp0 == 0
        self.cov_update, self.burn, self.tskip = o.cov_update, o.burn, o.tskip
        self.weights = (o.cfg.w_scam, o.cfg.w_am, o.cfg.w_de)
        self.ring = o.DE.copy()                       # physical ring image of the DE history
        self.head = 0
        al = torch.from_numpy
        self.t = dict(X=al(o.X), lnL=al(o.lnL), lp=al(o.lp), temp_of=al(o.temp_of), slot_of=al(o.slot_of),
                      Ut=al(o.Ut), S=al(o.S), cov=al(o.cov), DE=al(self.ring), AM=al(o.AM) if self.owns_cold else None)
        self.iter = 0
        self._eig_pending = False
        self.eig_lag = o.eig_lag                      # what ShardedPTEngine reads off its local engine

    def _eig_finish(self):
        if self._eig_pending:
            for w in range(self.o.Wc):
                self.o._svd(w)
        self._eig_pending = False

    def get(self, name):
        if name in ("nacc", "jstat", "nswap", "mu", "M2"):
            return getattr(self.o, name)
        return self.t[name].numpy()

    def init_state(self, p0):
        self.o.init_state(p0)

    def sync(self):
        pass

    def _logical_de(self):
        self.o.DE[...] = np.roll(self.ring, -self.head, axis=1)

    def mh_steps(self, iter0, nsteps):
        self._logical_de()
        err = orc.lib().orc_mh_steps(C.byref(self.o.cfg), C.byref(self.o._state()), iter0, nsteps, None)
        assert err == 0

    def gather_lnl(self, out):
        out.copy_(torch.from_numpy(np.ascontiguousarray(self.o.by_temp(self.o.lnL))))

    def sweep(self, it, lnl_glob, map_out):
        o = self.o
        lg = np.ascontiguousarray(lnl_glob.numpy())
        m = np.zeros((o.W, o.ntg), dtype=np.int32)
        acc = np.zeros((o.W, o.ntg), dtype=np.uint64)
        if o.swap_mode == "oddeven":
            orc.lib().orc_swap_oddeven(o.W, o.ntg, orc._p(o.ladder), orc._p(lg), it, o.seed, o.walker0,
                                       orc.swap_parity(it, o.tskip), orc._p(m, orc._ip), orc._p(acc, orc._up))
        else:
            err = orc.lib().orc_swap_sweep(o.W, o.ntg, orc._p(o.ladder), orc._p(lg), it, o.seed, o.walker0,
                                           orc._p(m, orc._ip), orc._p(acc, orc._up), None)
            assert err == 0
        o.nswap[:, o.temp0:o.temp0 + o.nt] += acc[:, o.temp0:o.temp0 + o.nt]
        map_out.copy_(torch.from_numpy(m))

    def write_am(self, it):
        o = self.o
        if self.owns_cold:
            o.AM[np.arange(o.W), it % o.cov_update] = o.X[np.arange(o.W), o.slot_of[:, 0]]
            o.AMflag[:, it % o.cov_update] = 2                     # the swap's row is a KEY row (AM row flags)

    def update_cov(self, it_done):
        o = self.o
        if o.per_walker:
            for w in range(o.W):
                o.cov[w] = orc.welford(o.AM[w], o.mu[w], o.M2[w], it_done)
        elif o.am_rle:
            o.cov[0] = orc.pool_update_rle(o.AM, o.AMflag, o.mu[0], o.M2[0], it_done)
        else:
            o.cov[0] = orc.pool_update(o.AM, o.mu[0], o.M2[0], it_done)
        if o.eig_lag:
            self._eig_pending = True                                # ShardedPTEngine.run finishes the epoch behind the next launch
            return
        for w in range(o.Wc):
            o._svd(w)

    def update_de(self):
        o = self.o
        mem = min(o.cov_update, o.burn)
        idx = (self.head + np.arange(mem)) % o.burn
        for wc in range(o.Wc):
            for r in range(mem):
                src_w = wc if o.per_walker else r % o.W
                self.ring[wc, idx[r]] = o.AM[src_w, o.cov_update - mem + r]
        self.head = (self.head + mem) % o.burn

    def set_de_head(self, head):
        self.head = head

    def set_de_active(self, on=True):
        self.o.cfg.de_on = int(on)
