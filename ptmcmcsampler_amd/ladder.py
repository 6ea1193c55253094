"""Temperature ladder of the reference (PTMCMCSampler/PTMCMCSampler.py:699-720)."""
import numpy as np


def temperature_ladder(nchain, ndim, Tmin=1, Tmax=None, tstep=None):
    """Geometric ladder; spacing ``1 + sqrt(2/ndim)`` unless ``Tmax`` fixes it.

    A single chain gets ``array([1])`` (integer dtype, hence the reference's file name
    ``chain_1.txt`` versus ``chain_1.0.txt`` for several chains)."""
    if nchain > 1:
        if tstep is None and Tmax is None:
            tstep = 1 + np.sqrt(2 / ndim)
        elif tstep is None and Tmax is not None:
            tstep = np.exp(np.log(Tmax / Tmin) / (nchain - 1))
        ladder = np.zeros(nchain)
        for ii in range(nchain):
            ladder[ii] = Tmin * tstep**ii
    else:
        ladder = np.array([1])
    return ladder
