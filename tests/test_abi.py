"""The C-ABI library loads and exports every symbol include/ptmi.h declares (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from ptmcmcsampler_amd import _lib
    if not os.path.exists(_lib.SO):
        ge.build()
    return _lib


def test_header_and_binding_list_the_same_symbols(lib):
    hdr = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ptmi_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    L = lib.load()
    for s in lib.SYMBOLS:
        assert hasattr(L, s), s
    assert L.ptmi_version() == 1


def test_struct_layouts_match_the_header(lib):
    import ctypes as C
    import subprocess
    import tempfile
    assert C.sizeof(lib.Config) == 26 * 4 + 2 * 8 + 8 + 8 + 2 * 8 + 4 * 8 + 3 * 8   # 26 int32, 2 doubles, seed, stream, pointers/lengths
    assert C.sizeof(lib.Buffers) == 22 * 8
    # the C compiler's view of include/ptmi.h
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "l.c")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "ptmi.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu", '
                             'sizeof(ptmi_config), offsetof(ptmi_config, swap_mode), offsetof(ptmi_config, seed), '
                             'offsetof(ptmi_config, group_mask), sizeof(ptmi_buffers), offsetof(ptmi_buffers, AMaux), '
                             'offsetof(ptmi_config, hmc_eps), offsetof(ptmi_config, gj_tab), offsetof(ptmi_buffers, gj), offsetof(ptmi_buffers, AMflag));return 0;}\n')
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(td, "l")])
        got = [int(v) for v in subprocess.check_output([os.path.join(td, "l")]).split()]
    assert got == [C.sizeof(lib.Config), lib.Config.swap_mode.offset, lib.Config.seed.offset, lib.Config.group_mask.offset,
                   C.sizeof(lib.Buffers), lib.Buffers.AMaux.offset, lib.Config.hmc_eps.offset, lib.Config.gj_tab.offset,
                   lib.Buffers.gj.offset, lib.Buffers.AMflag.offset]
    dims = (2, 32, 33, 100, 104, 105, 112, 113, 256, 416, 417, 512, 513, 640, 641, 1024, 1025, 2048)
    assert [lib.lanes_for(d) for d in dims] == [lib.load().ptmi_lanes_for(d) for d in dims]
    assert [lib.lanes_for(d, grad=True) for d in dims] == [lib.load().ptmi_lanes_for_grad(d) for d in dims]


def test_variant_flags_match_the_header(lib):
    """The PTMI_VAR_* flags the GPU tests assert on (which instantiation of the fused kernel ran) are the header's."""
    hdr = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    flags = {n: int(v) for n, v in re.findall(r"\bPTMI_(VAR_[A-Z_]+)\s*=\s*(\d+)", hdr)}
    high = {n: v for n, v in flags.items() if v >= 1 << 12}                                        # bits 12-27 carry the shape: a flag beyond the
    assert high == {"VAR_UTPAD": 1 << 28}                                                          # twelve low bits sits above them
    low = sorted(v for v in flags.values() if v < 1 << 12)
    assert len(low) >= 11 and low == [1 << k for k in range(len(low))]                             # distinct bits, none skipped
    for n, v in flags.items():
        assert getattr(lib, n) == v, n


def test_no_cpu_fallback(lib):
    """Without a device the product path refuses to run instead of computing on the host."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ptmcmcsampler_amd.engine import PTEngine
    with pytest.raises(lib.PtmiError):
        PTEngine(4, 2, 2, np.eye(4))
    assert lib.device_count() == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ptmcmcsampler_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("oracle/", "").lower() or f == "_lib.py" or "import oracle" not in src, f
                assert "from oracle" not in src and "import oracle" not in src, f


def test_create_rejects_bad_configurations(lib):
    """Argument validation of ptmi_create happens before any device is touched: error code and message, no crash."""
    import ctypes as C
    import numpy as np
    L = lib.load()
    d = 4
    ladder = np.array([1.0, 2.0])
    one = np.zeros(8)
    keep = dict(ladder=ladder, temps=ladder.copy(), tab=np.zeros(3 * d * d))

    def cfg(**over):
        kw = dict(ndim=d, ntemps=2, nwalkers=2, ntemps_global=2, w_scam=20, cov_update=10, de_size=10, tskip=10, cov_per_walker=1,
                  ladder=keep["ladder"].ctypes.data_as(lib._dp), temps_mh=keep["temps"].ctypes.data_as(lib._dp))
        kw.update(over)
        return lib.Config(**kw)

    buf = lib.Buffers(**{k: C.c_void_p(one.ctypes.data) for k in ("X", "lnL", "lp", "temp_of", "slot_of", "Ut", "S", "nacc", "jstat")})
    h = C.c_void_p()

    def err(c, b=buf):
        rc = L.ptmi_create(C.byref(c), C.byref(b), C.byref(h))
        assert rc != 0 and not h
        return rc, L.ptmi_last_error().decode()

    assert "No jump proposals" in err(cfg(w_scam=0))[1]
    assert "swap_mode" in err(cfg(swap_mode=7))[1]
    assert "outside ladder" in err(cfg(temp0=1))[1]
    assert "cov_update" in err(cfg(cov_update=0))[1]
    assert "whitening tables" in err(cfg(w_nuts=5))[1]
    tab = keep["tab"].ctypes.data_as(lib._dp)
    assert "gj buffer" in err(cfg(w_nuts=5, gj_tab=tab))[1]
    gbuf = lib.Buffers(**{k: C.c_void_p(one.ctypes.data) for k in ("X", "lnL", "lp", "temp_of", "slot_of", "Ut", "S", "nacc", "jstat", "gj")})
    rc, msg = err(cfg(ndim=600, w_nuts=5, gj_tab=tab), gbuf)
    assert rc == -3 and "ndim <= 512" in msg                                     # PTMI_EUNSUPPORTED
    assert "hmc_min" in err(cfg(w_hmc=5, gj_tab=tab, hmc_min=3, hmc_max=3), gbuf)[1]
    assert "required device buffer" in err(cfg(), lib.Buffers())[1]
    # a valid configuration gets as far as looking for a device
    rc, msg = err(cfg()) if lib.device_count() == 0 else (-4, "no HIP device")
    assert rc == -4 and "no HIP device" in msg


def test_product_ladder_matches_the_reference_fixture(lib, golden):
    """ptmi_temperature_ladder (host arithmetic behind the C ABI) against the reference's temperatureLadder outputs
    (PTMCMCSampler.py:699-720; fixture written by tests/golden/make_golden.py), bit for bit, incl. the integer lone chain."""
    import numpy as np
    from ptmcmcsampler_amd.ladder import temperature_ladder
    g = golden("ladder")
    for i, (n, d, Tmin, Tmax) in enumerate(g["cases"]):
        got = temperature_ladder(int(n), int(d), Tmin, None if Tmax < 0 else Tmax)
        assert np.array_equal(np.asarray(got, dtype=np.float64), g["ladder_%d" % i]), i
    one = temperature_ladder(1, 7)
    assert one.dtype.kind == "i" and "chain_{0}.txt".format(one[0]) == "chain_1.txt"      # PTMCMCSampler.py:285, :718
    assert "chain_{0}.txt".format(temperature_ladder(2, 2)[0]) == "chain_1.0.txt"
    assert np.allclose(temperature_ladder(4, 10, 1, None, 2.0), [1, 2, 4, 8])
