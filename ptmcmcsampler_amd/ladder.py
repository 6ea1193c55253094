"""Temperature ladder (the reference's ``temperatureLadder``, PTMCMCSampler/PTMCMCSampler.py:699-720).

The arithmetic lives behind the C ABI (``ptmi_temperature_ladder`` in ``csrc/ptmi_abi.hip``, host code);
this module only marshals the arguments."""
import ctypes as C

import numpy as np

from . import _lib


def temperature_ladder(nchain, ndim, Tmin=1, Tmax=None, tstep=None):
    """``nchain`` temperatures ``Tmin * tstep**i``.  Without ``Tmax`` / ``tstep`` the spacing is ``1 + sqrt(2/ndim)``.

    One chain returns the *integer* array ``[1]``: that dtype is why the reference names a lone chain's file
    ``chain_1.txt`` and a ladder's cold chain ``chain_1.0.txt`` (:285, :718)."""
    nchain = int(nchain)
    if nchain <= 1:
        return np.ones(1, dtype=np.int64)
    out = np.empty(nchain, dtype=np.float64)
    _lib.check(_lib.load().ptmi_temperature_ladder(
        nchain, int(ndim), float(Tmin), -1.0 if Tmax is None else float(Tmax), -1.0 if tstep is None else float(tstep),
        out.ctypes.data_as(C.POINTER(C.c_double))))
    return out
