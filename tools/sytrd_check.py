"""ptmi_eig_sytrd against numpy.linalg.eigh on a random covariance, and its time (developer tool, GPU):
   python tools/sytrd_check.py [ndim ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ptmcmcsampler_amd.engine import PTEngine
from ptmcmcsampler_amd import _lib
for d in [int(x) for x in sys.argv[1:]] or [1000, 300, 37]:
    g = PTEngine(d, 2, 2, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=100, burn=1000, tskip=10, seed=1, cov_mode="pooled",
                 use_de_buffer=False, eig_mode="sytrd")
    g.init_state(np.zeros(d))
    rng = np.random.default_rng(d)
    X = rng.standard_normal((4 * d, d)) * np.exp(rng.uniform(-1, 1, d))
    cov = X.T @ X / (4 * d)
    g.t["cov"][0].copy_(torch.from_numpy(cov))
    _lib.check(g.lib.ptmi_eig_sytrd(g.h, None, None, None)); g.sync()
    Ut, S = g.get("Ut")[0, 0], g.get("S")[0, 0]
    w = np.linalg.eigvalsh(cov)[::-1]
    rec = (Ut.T * S) @ Ut
    print("d=%d  eigenvalues %.2e  orthogonality %.2e  reconstruction %.2e (relative)" % (
        d, np.abs(S - w).max() / w.max(), np.abs(Ut @ Ut.T - np.eye(d)).max(), np.abs(rec - cov).max() / np.abs(cov).max()))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    for i in range(5):
        ev[i].record(); _lib.check(g.lib.ptmi_eig_sytrd(g.h, None, None, None))
    ev[5].record(); torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    c = g.t["cov"]
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.linalg.eigh(c); e0.record(); torch.linalg.eigh(c); e1.record(); torch.cuda.synchronize()
    print("   ptmi_eig_sytrd %.2f ms (min of %s)   torch.linalg.eigh %.2f ms" % (min(t), ["%.2f" % x for x in t], e0.elapsed_time(e1)))
