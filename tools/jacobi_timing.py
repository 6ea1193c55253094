"""ptmi_eig_jacobi alone at config-2 size (4096 covariances of 100 x 100 after one covariance epoch), and the sweeps the
same matrices take in the oracle's restatement (developer tool, GPU)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from ptmcmcsampler_amd import _lib
from ptmcmcsampler_amd.engine import PTEngine

d, nt, W = 100, 64, 4096
g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=1, cov_mode="per_walker",
             eig_mode="jacobi", use_de_buffer=False)
g.init_state(np.zeros(d))
g.run(1001)
g.sync()
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    _lib.check(g.lib.ptmi_eig_jacobi(g.h))
    torch.cuda.synchronize()
    print("ptmi_eig_jacobi, %d matrices: %.2f ms" % (W, (time.perf_counter() - t) * 1e3))
cov = g.get("cov")
sw = [orc.eig_jacobi(cov[w])[2] for w in (0, 1, 2, 100, 4095)]
print("sweeps (oracle, same matrices):", sw, " rounds per sweep:", d - 1)
S = g.get("S")[:, 0]
print("eigenvalue range of walker 0: %.3g .. %.3g" % (S[0].min(), S[0].max()))
