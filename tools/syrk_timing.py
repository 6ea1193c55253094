"""Timing of the pooled-statistics kernels on a filled AM buffer (developer tool, GPU):
   python tools/syrk_timing.py [ndim nwalkers ntemps reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ptmcmcsampler_amd.engine import PTEngine
from ptmcmcsampler_amd import _lib
a = [int(x) for x in sys.argv[1:]]
d, W, nt, reps = (a + [100, 4096, 8, 10][len(a):])[:4]
g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=1000, burn=100000, tskip=100, seed=1, cov_mode="pooled", use_de_buffer=False,
             eig_mode="lapack")
g.init_state(np.zeros(d)); g.run(2000); g.sync()
fl = g.get("AMflag")
print("stored share %.3f" % float((fl & 3 != 0).mean()))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
_lib.check(g.lib.ptmi_update_cov(g.h, 3000))
ev[0].record()
for i in range(reps):
    _lib.check(g.lib.ptmi_update_cov(g.h, 3000)); ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
print("ptmi_update_cov (rle list + syrk + reduce + finish): min %.3f  median %.3f ms" % (min(t), sorted(t)[len(t) // 2]))
