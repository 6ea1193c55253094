"""Temperature-block sharding of one ladder over the GPUs of a node.

Rank g of the process group owns the temperature ranks ``[g*nt, (g+1)*nt)`` of every
walker (state ``[W][nt][d]`` on its GPU).  Chains only interact at

* swap epochs (every Tskip, PTMCMCSampler/PTMCMCSampler.py:631-697): lnL is
  all-gathered (``W * ntemps * 8`` bytes), every GPU runs the identical hot->cold sweep
  (same Philox uniforms, same data -> same map), and only rows whose new position is in
  another block travel: by grouped send/recv with the two neighbouring blocks (RCCL over one xGMI
  link each way), and by an all-to-all only in the rare epoch where a row crosses a whole block;
* covariance / DE epochs (:545-576): the GPU holding rank 0 adapts and broadcasts the
  factorization ``Ut, S`` (not the covariance: only the owner reads it) and the new DE rows.

The exchange below is written with torch tensor ops so it runs unchanged on CUDA tensors
over RCCL ("nccl") and on CPU tensors over gloo (the CPU tests drive it with an
oracle-backed stand-in for the per-GPU kernels).
"""
import numpy as np


def _torch():
    import torch
    return torch


def plan_exchange(map_glob, slot_of, temp0, nt, rank, world):
    """Who sends which row where, from the sweep's global map.

    ``map_glob[w][j]`` = position whose state moves to position j (all ranks of the
    ladder).  Returns a dict of tensors describing this rank's part; every index list is
    in row-major (walker, position) order, which is what makes sender and receiver agree
    without exchanging any metadata.
    """
    torch = _torch()
    W, ntg = map_glob.shape
    dev = map_glob.device
    m = map_glob.long()
    inv = torch.empty_like(m)
    inv.scatter_(1, m, torch.arange(ntg, device=dev).expand(W, ntg))        # inv[w][map[w][j]] = j
    src_pos = m[:, temp0:temp0 + nt]                                         # source of each local destination
    dst_pos = inv[:, temp0:temp0 + nt]                                       # destination of each local source
    src_rank = torch.div(src_pos, nt, rounding_mode="floor")
    dst_rank = torch.div(dst_pos, nt, rounding_mode="floor")
    so = slot_of.long()
    # local moves: new_slot[w][j] = old slot of the local source
    local_dst = src_rank == rank
    new_slot = torch.zeros_like(so)
    loc_src = (src_pos - temp0).clamp(0, nt - 1)
    new_slot[local_dst] = torch.gather(so, 1, loc_src)[local_dst]
    # rows that leave / arrive.  Three host synchronisations in all (two nonzero, one count read-back); the index
    # lists only hold the rows that cross a block edge.
    leaving = dst_rank != rank
    lw, lp_ = torch.nonzero(leaving, as_tuple=True)                          # row-major (w, local source position)
    free_slots = so[lw, lp_]                                                 # k-th leaving row of a walker frees the k-th slot
    arriving = ~local_dst
    aw, aj = torch.nonzero(arriving, as_tuple=True)                          # row-major (w, local destination)
    assert free_slots.numel() == aw.numel()
    new_slot[aw, aj] = free_slots                                            # per walker the counts match, so the lists align
    lq, ld = dst_rank[lw, lp_], dst_pos[lw, lp_]
    order = torch.argsort(lq * (W * ntg) + lw * ntg + ld)                    # by destination rank, then the receiver's (walker, position)
    send_idx = torch.stack([lw[order], free_slots[order]], 1)
    aq = src_rank[aw, aj]
    rorder = torch.argsort(aq * (W * nt) + aw * nt + aj)                     # by source rank, then (walker, position)
    counts = torch.zeros((2, world), dtype=torch.long, device=dev)
    counts[0].scatter_add_(0, lq, torch.ones_like(lq))
    counts[1].scatter_add_(0, aq, torch.ones_like(aq))
    counts = counts.cpu().tolist()
    return dict(new_slot=new_slot, send_idx=send_idx, send_counts=counts[0],
                recv_w=aw[rorder], recv_j=aj[rorder], recv_counts=counts[1])


class DistComm(object):
    """The three collectives the sharded engine needs, over torch.distributed ("nccl" = RCCL, or gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.root = dist.get_global_rank(group, 0) if group is not None else 0

    def all_gather(self, out, inp):
        """out[q] <- rank q's inp.  xGMI is point to point with a link to EVERY peer of the node (7 per GPU): rank r sends its block
        to all peers at once, one link each, in ONE grouped send/recv (ncclSend / ncclRecv inside a group) -- a single hop of
        W x ntemps x 8 bytes whatever the world size, where a ring all-gather takes world - 1 hops in turn (the swap epoch's lnL
        gather at 8 GPUs: 7 x 33 us modelled, tools/scale_model.py).  PTMI_ALLGATHER=ring keeps the library's collective."""
        import os
        dist = self.dist
        if self.world <= 2 or os.environ.get("PTMI_ALLGATHER") == "ring":
            dist.all_gather_into_tensor(out, inp, group=self.group)
            return
        parts = out.view((self.world,) + tuple(inp.shape)) if out.shape != (self.world,) + tuple(inp.shape) else out
        gr = (lambda q: dist.get_global_rank(self.group, q)) if self.group is not None else (lambda q: q)
        staged = inp.is_cuda and dist.get_backend(self.group) != "nccl"          # gloo rehearsals move device tensors through the host
        src = inp.cpu() if staged else inp.contiguous()
        bufs = {q: (src.new_empty(src.shape) if staged else parts[q]) for q in range(self.world) if q != self.rank}
        ops = []
        for q in bufs:
            ops.append(dist.P2POp(dist.isend, src, gr(q), group=self.group))
            ops.append(dist.P2POp(dist.irecv, bufs[q], gr(q), group=self.group))
        parts[self.rank].copy_(inp)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if staged:
            for q, b in bufs.items():
                parts[q].copy_(b)

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        self.dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    def broadcast(self, t):
        self.dist.broadcast(t, src=self.root, group=self.group)

    def neighbour_exchange(self, send, recv):
        """send[q] -> rank q and recv[q] <- rank q for q = rank - 1, rank + 1 only: one grouped send/recv pair per block
        edge (RCCL: ncclSend / ncclRecv inside one group, i.e. one xGMI link each way).  Over gloo (CPU tests, one-GPU
        rehearsals) device tensors are staged through the host."""
        import torch
        dist = self.dist
        peers = [q for q in (self.rank - 1, self.rank + 1) if 0 <= q < self.world]
        if not peers:
            return
        gr = (lambda q: dist.get_global_rank(self.group, q)) if self.group is not None else (lambda q: q)
        staged = send.is_cuda and dist.get_backend(self.group) != "nccl"
        sbuf = {q: (send[q].cpu() if staged else send[q]) for q in peers}
        rbuf = {q: (torch.empty_like(sbuf[q]) if staged else recv[q]) for q in peers}
        ops = []
        for q in peers:
            ops.append(dist.P2POp(dist.isend, sbuf[q], gr(q), group=self.group))
            ops.append(dist.P2POp(dist.irecv, rbuf[q], gr(q), group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if staged:
            for q in peers:
                recv[q].copy_(rbuf[q])


class ShardedPTEngine(object):
    """One ladder of ``ntemps_global`` ranks sharded over ``group``; same interface as PTEngine."""

    def __init__(self, ndim, ntemps_global, nwalkers, cov0, group=None, local_factory=None, comm=None, **kw):
        torch = _torch()
        self.comm = comm if comm is not None else DistComm(group)
        self.rank, self.world = self.comm.rank, self.comm.world
        if ntemps_global % self.world:
            raise ValueError("ntemps_global=%d is not a multiple of the %d ranks" % (ntemps_global, self.world))
        self.ntg, self.nt = ntemps_global, ntemps_global // self.world
        self.temp0 = self.rank * self.nt
        self.d, self.W = ndim, nwalkers
        if local_factory is None:
            from .engine import PTEngine
            local_factory = PTEngine
        if int(kw.get("eig_lag", 0)) > 0 and kw.get("eig_mode") == "ql" and kw.get("cov_mode", "per_walker") == "per_walker":
            # (PTEngine: one GPU only -- every block would factorize its own walkers' tables late, with nothing to order the tables' arrival)
            raise ValueError("eig_lag with eig_mode='ql' and per-walker covariances is a one-GPU mode; a sharded ladder takes it with the pooled covariance")
        self.local = local_factory(ndim, self.nt, nwalkers, cov0, ntemps_global=ntemps_global, temp0=self.temp0, **kw)
        L = self.local
        self.t, self.owns_cold, self.device = L.t, L.owns_cold, L.device
        self.cov_update, self.burn, self.tskip, self.weights = L.cov_update, L.burn, L.tskip, L.weights
        self.stream = getattr(L, "stream", None)
        self.iter, self.swap_proposed = 0, 0
        self.de_head = 0
        # eig_lag = L (PTEngine): the owner of rank 0 factorizes the pooled covariance while every GPU runs the L launches that follow
        # the epoch -- on the host (eig_mode "lapack"), or on its side stream (the device factorizations: "sytrd", "hipsolver"; with
        # stats_async the statistics too) -- and the table is broadcast behind the L-th launch's swap: nobody waits at the broadcast
        # for GPU 0's statistics and factorization.  The same decision on every rank (from the configuration alone).
        self.eig_lag = int(getattr(L, "eig_lag", 0))                       # PTEngine: 0 where the late table does not apply
        self._bcast_pending, self._bcast_wait = False, 0
        # stats_async (PTEngine): the owner switches rings at a covariance epoch, so a DE epoch that falls on it goes first -- on EVERY
        # rank (the epochs' broadcasts are collectives: one order for all), hence from the configuration, not from the local engine
        self._de_first = bool(kw.get("stats_async", False)) and self.eig_lag >= 1
        self._late_finish = bool(getattr(L, "late_finish", False))           # (PTEngine derives it from the configuration: the same on every rank)
        self._lnl_loc = torch.zeros((self.W, self.nt), dtype=torch.float64, device=self.device)
        self._map = torch.zeros((self.W, self.ntg), dtype=torch.int32, device=self.device)
        self._parts = torch.empty((self.world * self.W, self.nt), dtype=torch.float64, device=self.device)
        self.rows_moved = 0
        self.neighbour_swaps = 0                                                 # swap epochs served by the two neighbour links alone
        # device-side exchange (HIP engines): fixed [world][W][d+2] buffers; the plan, the packing and the apply step need no host
        # round trip -- the ONE host wait per swap epoch is for the 4-byte multi-hop flag (an event behind the pack step, not a stream
        # synchronisation: the neighbour exchange is already queued when the host waits, swap())
        self.device_exchange = hasattr(L, "exchange_pack")
        if self.device_exchange:
            self._send = torch.zeros((self.world, self.W, self.d + 2), dtype=torch.float64, device=self.device)
            self._recv = torch.zeros_like(self._send)

    # delegation
    def get(self, name):
        return self.local.get(name)

    def am_params(self, rows):
        return self.local.am_params(rows) if hasattr(self.local, "am_params") else rows

    @property
    def am_rle(self):
        return bool(getattr(self.local, "am_rle", False))

    def am_expand(self, *args, **kw):
        if hasattr(self.local, "am_expand"):
            self.local.am_expand(*args, **kw)

    def am_period(self, it=None):
        return self.local.am_period(self.iter if it is None else it)

    def init_state(self, p0):
        p0 = np.asarray(p0, dtype=np.float64)
        if p0.ndim == 3:                                                      # [W][ntemps_global][d] -> my block
            p0 = p0[:, self.temp0:self.temp0 + self.nt]
        self.local.init_state(p0)

    def mh_steps(self, iter0, nsteps):
        self.local.mh_steps(iter0, nsteps)

    def sync(self):
        self.local.sync()

    # ---- swap epoch ----------------------------------------------------------------------
    def swap(self, it):
        torch, L = _torch(), self.local
        W, nt, ntg, d = self.W, self.nt, self.ntg, self.d
        L.gather_lnl(self._lnl_loc)
        self.comm.all_gather(self._parts, self._lnl_loc)                      # [world][W][nt]
        if self.device_exchange:
            L.sweep_blocks(it, self._parts, self._map)                        # identical on every rank
            L.exchange_pack(self._map, self._send)                            # tables rewritten, leaving rows packed
            # A sweep moves a walker's rows between NEIGHBOURING blocks unless its carried state wins every pair of a
            # whole block: then (and only then) the all-to-all is needed; every rank reads the same flag off the map.
            # The neighbour exchange is queued BEFORE the flag is read: every epoch needs it, and the four bytes (requested
            # behind the pack step with their own event) arrive while it runs -- the host decides with the GPU still busy,
            # and the launches that follow are queued before the stream runs dry.  In the rare multi-hop epoch the
            # all-to-all then delivers every segment, the two edge ones again (same rows).
            neighbour = hasattr(self.comm, "neighbour_exchange")
            if neighbour:
                self.comm.neighbour_exchange(self._send, self._recv)
            if neighbour and not L.exchange_multihop():
                self.neighbour_swaps += 1
            else:
                self.comm.all_to_all(self._recv, self._send)                  # equal splits: send[q] -> rank q
            L.exchange_apply(self._recv)
            L.write_am(it)
            self.swap_proposed += 1
            return
        lnl_glob = self._parts.view(self.world, W, nt).permute(1, 0, 2).reshape(W, ntg).contiguous()
        L.sweep(it, lnl_glob, self._map)                                      # identical on every rank
        plan = plan_exchange(self._map, L.t["slot_of"], self.temp0, nt, self.rank, self.world)
        X, lnL, lp = L.t["X"], L.t["lnL"], L.t["lp"]
        si = plan["send_idx"]
        send = torch.cat([X[si[:, 0], si[:, 1]], lnL[si[:, 0], si[:, 1]].unsqueeze(1), lp[si[:, 0], si[:, 1]].unsqueeze(1)], 1) \
            if si.numel() else torch.zeros((0, d + 2), dtype=torch.float64, device=self.device)
        nrecv = int(sum(plan["recv_counts"]))
        recv = torch.empty((nrecv, d + 2), dtype=torch.float64, device=self.device)
        self.comm.all_to_all(recv, send.contiguous(), plan["recv_counts"], plan["send_counts"])
        new_slot = plan["new_slot"]
        if nrecv:
            rw, rj = plan["recv_w"], plan["recv_j"]
            rs = new_slot[rw, rj]
            X[rw, rs] = recv[:, :d]
            lnL[rw, rs] = recv[:, d]
            lp[rw, rs] = recv[:, d + 1]
        L.t["slot_of"].copy_(new_slot.to(L.t["slot_of"].dtype))
        pos = torch.arange(nt, device=self.device, dtype=L.t["temp_of"].dtype).expand(W, nt).contiguous()
        L.t["temp_of"].scatter_(1, new_slot, pos)
        L.write_am(it)
        self.swap_proposed += 1
        self.rows_moved += nrecv

    # ---- covariance / DE epochs (PTMCMCSampler.py:545-576) ----------------------------------
    def update_cov(self, it_done):
        L = self.local
        # a table still pending from the epoch before takes effect here at the latest: before this epoch's statistics -- or, with a
        # device factorization on the owner's side stream (PTEngine.late_finish), behind them: the owner's update_cov then runs
        # statistics, finishes the old table, starts the new factorization, and the old table is broadcast after it
        late = self._bcast_pending and self._late_finish
        if self._bcast_pending and not late:
            self._finish_table()
        if self.owns_cold:
            L.update_cov(it_done)
        if late:
            self._bcast_table()
            self._bcast_pending, self._bcast_wait = False, 0
        if self.eig_lag:
            self._bcast_pending, self._bcast_wait = True, self.eig_lag    # run() finishes the epoch eig_lag launches later
            return
        self._bcast_table()

    def _finish_table(self):
        if self.owns_cold and getattr(self.local, "_eig_pending", False):
            self.local._eig_finish() if hasattr(self.local, "_eig_finish") else self.local._eig_end()
        self._bcast_table()
        self._bcast_pending, self._bcast_wait = False, 0

    def _bcast_table(self):
        # the other blocks only ever read the factorization: the covariance itself stays where it is adapted
        # (pooled: 80 KB of eigenvectors per epoch at ndim = 100; per walker: a third less than with cov)
        L = self.local
        for name in ("Ut", "S"):
            self.comm.broadcast(L.t[name])

    def update_de(self, it_done=None):
        torch, L = _torch(), self.local
        if L.t.get("DE") is None:
            return
        size, mem = self.burn, min(self.cov_update, self.burn)
        idx = (self.de_head + torch.arange(mem, device=self.device)) % size
        if self.owns_cold:
            if it_done is not None and getattr(L, "am_rle", False):
                L.update_de(it_done)                                  # every row of the ring first (PTEngine.am_expand)
            else:
                L.update_de()
            rows = L.t["DE"][:, idx].contiguous()
        else:
            rows = torch.empty((L.t["DE"].shape[0], mem, L.t["DE"].shape[2]), dtype=torch.float64, device=self.device)
        self.comm.broadcast(rows)
        if not self.owns_cold:
            L.t["DE"][:, idx] = rows
            L.set_de_head((self.de_head + mem) % size)
        self.de_head = (self.de_head + mem) % size

    def _epochs(self, it):
        cu, burn = self.cov_update, self.burn
        de_now = (it - 1) % burn == 0 and it - 1 != 0
        de_first = self._de_first                  # the owner's DE epoch reads the finished period's ring: before its switch
        if de_now and de_first:
            self.update_de(it - 1)
        if (it - 1) % cu == 0 and it - 1 != 0:
            self.update_cov(it - 1)
        if de_now and not de_first:
            self.update_de(it - 1)
        if it - 1 == burn and self.weights[2] > 0 and self.local.t.get("DE") is not None:
            self.local.set_de_active(True)

    def run(self, niter):
        last = self.iter + niter
        it = self.iter + 1
        while it <= last:
            self._epochs(it)
            end = last
            for per in (self.cov_update, self.burn) + ((self.tskip,) if self.tskip > 0 and self.ntg > 1 else ()):
                end = min(end, ((it - 1) // per + 1) * per)
            self.mh_steps(it, end - it + 1)
            if self.tskip > 0 and self.ntg > 1 and end % self.tskip == 0:
                self.swap(end)
            if self._bcast_pending:                                       # the owner factorized while those launches ran
                self._bcast_wait -= 1
                # (when the next iteration opens a covariance epoch and the owner finishes behind its statistics: update_cov does it)
                if self._bcast_wait <= 0 and not (self._late_finish and end % self.cov_update == 0 and end + 1 <= last):
                    self._finish_table()
            it = end + 1
        self.iter = last
        self.local.iter = last
