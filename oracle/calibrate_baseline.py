#!/usr/bin/env python
"""Times oracle/numpy_port.py next to the real reference (build container only) and writes
oracle/baseline_calibration.json.  Same workload on both: one chain, 100-d iso-Gaussian,
flat prior, cov0 = 0.01 I, covUpdate=1000, thin=10, SCAM-only and the default mix."""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import numpy_port as npp  # noqa: E402


def ref_rate(ndim, niter, weights):
    sys.path.insert(0, "/root/reference")
    ver = types.ModuleType("PTMCMCSampler.version")
    ver.version = "0+ref"
    sys.modules["PTMCMCSampler.version"] = ver
    from PTMCMCSampler import PTMCMCSampler as PT
    s = PT.PTSampler(ndim, npp.iso_logl, npp.flat_logp, np.eye(ndim) * 0.01, outDir=tempfile.mkdtemp(), verbose=False, seed=7)
    t0 = time.perf_counter()
    s.sample(np.zeros(ndim), niter, covUpdate=1000, burn=10000, thin=10, isave=1000,
             SCAMweight=weights[0], AMweight=weights[1], DEweight=weights[2])
    return niter / (time.perf_counter() - t0)


def port_rate(ndim, niter, weights):
    c = npp.ChainPort(ndim, npp.iso_logl, npp.flat_logp, np.eye(ndim) * 0.01, 1.0, 1000, 10000, weights, 7)
    t0 = time.perf_counter()
    c.run(np.zeros(ndim), niter)
    return niter / (time.perf_counter() - t0)


out = {"host": "build container, 1 core", "cases": []}
for name, w, n in (("scam_only", (20, 0, 0), 12000), ("default_mix", (20, 20, 20), 12000)):
    r, p = ref_rate(100, n, w), port_rate(100, n, w)
    out["cases"].append(dict(name=name, ndim=100, niter=n, reference_updates_per_s=r, port_updates_per_s=p, ratio=p / r))
    print(name, r, p, p / r)
json.dump(out, open(os.path.join(HERE, "baseline_calibration.json"), "w"), indent=1)
