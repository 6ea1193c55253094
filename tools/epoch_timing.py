"""Host-side timing of the covariance epoch pieces (developer tool, GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ptmcmcsampler_amd.engine import PTEngine, _blas_single_thread
from ptmcmcsampler_amd import _lib
d, nt, W = 100, 64, 4096
g = PTEngine(d, nt, W, np.eye(d) * 0.01, weights=(20, 0, 0), cov_update=1000, burn=10000, tskip=100, seed=1, cov_mode="pooled", use_de_buffer=False)
g.init_state(np.zeros(d)); g.run(1000); g.sync()
def T(f, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("ptmi_update_cov kernels ms", T(lambda: _lib.check(g.lib.ptmi_update_cov(g.h, 1000))))
print("get cov ms", T(lambda: g.get("cov")))
cov = g.get("cov")[0]
print("svd default threads ms", T(lambda: np.linalg.svd(cov)))
def s1():
    with _blas_single_thread(): np.linalg.svd(cov)
print("svd 1 thread ms", T(s1))
print("eig_host (svd+upload) ms", T(lambda: g._eig_host(0, cov)))
print("full update_cov ms", T(lambda: g.update_cov(1000)))
print("mh 100 steps ms", T(lambda: g.mh_steps(1001, 100)))
print("swap ms", T(lambda: g.swap(1100)))
