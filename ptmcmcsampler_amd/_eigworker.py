"""Host worker for the per-walker covariance epochs: LAPACK SVDs of a chunk of walkers (PTMCMCSampler.py:145, 803).

Runs in plain spawned interpreters (numpy only, no GPU state), each call single-threaded, so every walker's
factorization has the bits of a lone reference run whatever the number of workers.  numpy's svd does not scale over
Python threads, hence processes."""
import numpy as np


def _single_thread():
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1)
    except ImportError:
        import contextlib
        return contextlib.nullcontext()


def svd_chunk(args):
    """args = (covs [n][d][d], groups or None) -> (Ut [n][Ng][d][d], S [n][Ng][d]) embedded like PTEngine.put_eig."""
    covs, groups = args
    n, d = covs.shape[0], covs.shape[1]
    whole = groups is None
    if whole:
        groups = [np.arange(d)]
    Ut = np.zeros((n, len(groups), d, d))
    Sv = np.zeros((n, len(groups), d))
    with _single_thread():
        for w in range(n):
            for gi, g in enumerate(groups):
                U, S, _ = np.linalg.svd(covs[w] if whole else covs[w][np.ix_(g, g)])
                Ut[w, gi][np.ix_(np.arange(len(g)), g)] = U.T
                Sv[w, gi, :len(g)] = S
    return Ut, Sv
