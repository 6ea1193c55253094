"""The reference's RECORDED draws fed straight into the HIP kernels (``ptmi_test_replay``, include/ptmi.h) and compared with the
reference's recorded results -- no oracle in between.  The fixtures (tests/golden/make_golden.py) hold, for PTswap
(PTMCMCSampler.py:666-686): ladder, likelihoods, the n - 1 uniforms rank 0 drew and the resulting permutation / credits, for
ladders of 2 ... 512 ranks; for the SCAM and DE proposals (:820-876, :936-985): the point, the eigenvectors / DE history, every
draw ``sampler.stream`` handed out and the proposed point q.

Run on the GPU box: ``python -m pytest tests -m gpu``.  Nothing here reads /root/reference."""
import ctypes as C
import os

import numpy as np
import pytest

from test_gpu_parity import assert_same, mods  # noqa: F401  (mods is a fixture)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _word(k, n):
    """The smallest 32-bit word h with (h * n) >> 32 == k: what the kernels' index draw must see to return k."""
    return ((int(k) << 32) + int(n) - 1) // int(n)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_swap_sweep_on_the_references_uniforms(mods, fused, monkeypatch):
    """swap_fused_kernel (and the prepare + sweep pair) with the uniforms of the reference's own PTswap calls: the permutation
    of the likelihoods, of the states and the credits to the lower rank are the reference's, for n = 2, 4, 8, 16, 64, 512."""
    import torch
    orc, _lib, PTEngine = mods
    monkeypatch.setenv("PTMI_SWAP_FUSED", fused)
    g = np.load(os.path.join(GOLD, "ptswap.npz"))
    seen = set()
    for ci, (n, dd) in enumerate(g["meta"]):
        n = int(n)
        lnL, ladder, u = g["lnL_%d" % ci], g["ladder_%d" % ci], g["u_%d" % ci]
        e = PTEngine(2, n, 1, np.eye(2), ladder=ladder, weights=(20, 0, 0), tskip=1, cov_mode="pooled")
        e.init_state(np.zeros(2))
        e.put("lnL", lnL[None])                                     # slot s holds rank s at the start
        u_k = np.array(u[::-1], dtype=np.float64)        # drawn hottest pair first (k = n - 2 ... 0): index by k
        ud = torch.from_numpy(u_k).to(e.device)
        _lib.check(e.lib.ptmi_test_replay(e.h, C.c_void_p(ud.data_ptr()), None))
        e.swap(1)
        e.sync()
        so = e.get("slot_of")[0].astype(np.int64)                    # slot (= starting rank) whose state now sits at each rank
        assert np.array_equal(lnL[so], g["newlnL_%d" % ci], equal_nan=True), (ci, n)
        assert np.array_equal(g["p0s_%d" % ci][so], g["newp0s_%d" % ci]), (ci, n)
        assert np.array_equal(e.get("nswap")[0].astype(float), g["acc_%d" % ci]), (ci, n)
        seen.add(n)
    assert {2, 4, 64, 512} <= seen


@pytest.mark.parametrize("am_in_cycle", [True, False])
def test_scam_and_de_proposals_on_the_references_draws(mods, am_in_cycle):
    """The production split-path kernels -- propose_kernel (propose() of csrc/ptmi_mh.inc.h: cycles with AM entries) and the row
    kernel of csrc/ptmi_split.hip (cycles without) -- with the reference's recorded group, scale-branch, direction / row and
    normal / scale draws: q is the reference's q, bit for bit, at T = 1, 3.7, 100, 101 (no sqrt(T) scaling above 100, :861-862)
    and 1e80, for ndim = 5, 20, 100."""
    import torch
    orc, _lib, PTEngine = mods
    g = np.load(os.path.join(GOLD, "proposals.npz"))
    meta = g["meta"]
    T97, T90 = int(0.97 * 4294967296.0), int(0.9 * 4294967296.0)
    groups = {}
    for ci, (d, temp, kind, qxy) in enumerate(meta):
        if int(kind) in (0, 2):                                     # SCAM, DE (AM: BLAS summation order, covered through the oracle)
            groups.setdefault((int(d), float(temp)), []).append(ci)
    checked = {0: 0, 2: 0}
    for (d, temp), cases in groups.items():
        W = len(cases)
        Bn = g["DE_d%d" % d].shape[0]
        e = PTEngine(d, 1, W, np.eye(d), ladder=[temp], weights=(1, 1, 1) if am_in_cycle else (1, 0, 1), cov_update=4, burn=Bn, tskip=0, split=True,
                     cov_mode="pooled")
        e.put_eig(g["U_d%d" % d], g["S_d%d" % d])
        DE = np.zeros((1, Bn, e.de_ld))
        if e.de_epl:
            lane, slot = np.arange(d) % 4, np.arange(d) // 4
            DE[0][:, 8 * (slot // 2) + 2 * lane + slot % 2] = g["DE_d%d" % d]
        else:
            DE[0] = g["DE_d%d" % d]
        e.t["DE"].copy_(torch.from_numpy(DE))
        e.set_de_active(True)
        e.init_state(np.stack([g["x_%d" % ci] for ci in cases])[:, None, :])
        words = np.zeros((W, 1, 4), dtype=np.uint64)
        for w, ci in enumerate(cases):
            kind = int(meta[ci][2])
            dk, dv, db = g["dk_%d" % ci], g["dv_%d" % ci], g["db_%d" % ci]
            assert dk[0] == 0 and db[0] == 1                        # the group draw (one group)
            pick = _word(kind, 3) if am_in_cycle else _word(kind // 2, 2)    # the cycle entry: SCAM | AM | DE, or SCAM | DE
            if kind == 0:
                assert list(dk) == [0, 1, 0, 2] and db[2] == d
                prob, k, z = dv[1], int(dv[2]), dv[3]
                plo = 0xFFFFFFFF if prob > 0.97 else (T97 if prob > 0.9 else 0)      # PT:846-858 on the recorded uniform
                words[w, 0] = [(pick << 32) | plo, 0, _word(k, d) << 32, np.float64(z).view(np.uint64)]
            else:
                ints = [int(v) for kk, v in zip(dk[1:], dv[1:]) if kk == 0]
                unis = [v for kk, v in zip(dk[1:], dv[1:]) if kk == 1]
                mm, nn = ints[0], ints[-1]                           # PT:961-966: redrawn until the rows differ
                assert mm != nn and all(v == mm for v in ints[1:-1])
                prob = unis[0]
                plo = 0xFFFFFFFF if prob > 0.5 else 0
                rr = unis[1] if prob <= 0.5 else 0.0                 # PT:976: the scale uniform, handed over as a double
                off = (nn - mm - 1) % Bn
                words[w, 0] = [(pick << 32) | plo, (_word(mm, Bn) << 32) | _word(off, Bn - 1), 0, np.float64(rr).view(np.uint64)]
        wd = torch.from_numpy(words.view(np.int64)).to(e.device)
        _lib.check(e.lib.ptmi_test_replay(e.h, None, C.c_void_p(wd.data_ptr())))
        _lib.check(e.lib.ptmi_propose(e.h, 1))
        e.sync()
        Q, qaux = e.t["Q"].cpu().numpy(), e.t["qaux"].cpu().numpy()
        for w, ci in enumerate(cases):
            kind = int(meta[ci][2])
            assert int(qaux[w, 0, 1]) == kind and qaux[w, 0, 0] == 0.0
            assert_same(Q[w, 0], g["q_%d" % ci], "case %d (d=%d, T=%g, kind %d)" % (ci, d, temp, kind))
            checked[kind] += 1
    assert checked[0] >= 100 and checked[2] >= 100
