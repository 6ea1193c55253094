"""Stage-by-stage parity diagnosis on the GPU (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from ptmcmcsampler_amd.engine import PTEngine
from ptmcmcsampler_amd import _lib

def same(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return bool(np.array_equal(a.view(np.uint8), b.view(np.uint8)))

for d in [int(v) for v in sys.argv[1:]] or [100, 300, 1000]:
    rs = np.random.RandomState(d)
    A = rs.randn(d, d); cov0 = (A @ A.T / d + 0.5 * np.eye(d)) * 0.01
    p0 = rs.randn(3, 2, d) * 0.3
    for name, w in (("scam", (5, 0, 0)), ("am", (0, 5, 0)), ("de", (1, 0, 50))):
        kw = dict(weights=w, cov_update=50, burn=20, tskip=0, seed=7)
        g = PTEngine(d, 2, 3, cov0, **kw); o = orc.OracleEngine(d, 2, 3, cov0, **kw)
        g.init_state(p0); o.init_state(p0)
        if name == "de":
            de = rs.randn(3, 20, d); g.put("DE", de); o.DE[...] = de
            g.set_de_active(True); o.cfg.de_on = 1
        g.mh_steps(1, 15); err = orc.lib().orc_mh_steps(orc.C.byref(o.cfg), orc.C.byref(o._state()), 1, 15, None)
        g.sync()
        print(d, name, "X", same(g.get("X"), o.X), "lnL", same(g.get("lnL"), o.lnL), "nacc", same(g.get("nacc"), o.nacc), "AM", same(g.get("AM"), o.AM))
    # welford alone
    g = PTEngine(d, 1, 2, cov0, weights=(1, 0, 0), cov_update=50, burn=50, tskip=0)
    am = rs.randn(2, 50, d); g.put("AM", am)
    mu = np.zeros((2, d)); M2 = np.zeros((2, d, d))
    for ep in range(2):
        _lib.check(g.lib.ptmi_update_cov(g.h, (ep + 1) * 50)); g.sync()
        cov = [orc.welford(am[w], mu[w], M2[w], (ep + 1) * 50) for w in range(2)]
        print(d, "welford ep", ep, "mu", same(g.get("mu"), mu), "M2", same(g.get("M2"), M2), "cov", same(g.get("cov"), np.stack(cov)))
    # where does X differ?
    kw = dict(weights=(5, 0, 0), cov_update=50, burn=20, tskip=0, seed=7)
    g = PTEngine(d, 2, 3, cov0, **kw); o = orc.OracleEngine(d, 2, 3, cov0, **kw)
    g.init_state(p0); o.init_state(p0)
    print("init X same", same(g.get("X"), o.X), "lnL", same(g.get("lnL"), o.lnL))
    g.mh_steps(1, 1); orc.lib().orc_mh_steps(orc.C.byref(o.cfg), orc.C.byref(o._state()), 1, 1, None); g.sync()
    gx, ox = g.get("X"), o.X
    bad = np.argwhere(gx != ox)
    print("after 1 step: nbad", len(bad), "nacc", g.get("nacc").ravel(), o.nacc.ravel())
    if len(bad):
        print(" first", bad[:5].tolist(), " last", bad[-3:].tolist(), "elements idx set", sorted(set(bad[:, 2] % 16))[:20], sorted(set(bad[:, 2] // 16))[:30])
        w, s, i = bad[0]
        print(" vals", gx[w, s, i], ox[w, s, i], p0[w, s, i])
