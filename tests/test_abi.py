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
    assert C.sizeof(lib.Buffers) == 19 * 8
    # the C compiler's view of include/ptmi.h
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "l.c")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "ptmi.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu", '
                             'sizeof(ptmi_config), offsetof(ptmi_config, swap_mode), offsetof(ptmi_config, seed), '
                             'offsetof(ptmi_config, group_mask), sizeof(ptmi_buffers), offsetof(ptmi_buffers, AMaux), '
                             'offsetof(ptmi_config, hmc_eps), offsetof(ptmi_config, gj_tab), offsetof(ptmi_buffers, gj));return 0;}\n')
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(td, "l")])
        got = [int(v) for v in subprocess.check_output([os.path.join(td, "l")]).split()]
    assert got == [C.sizeof(lib.Config), lib.Config.swap_mode.offset, lib.Config.seed.offset, lib.Config.group_mask.offset,
                   C.sizeof(lib.Buffers), lib.Buffers.AMaux.offset, lib.Config.hmc_eps.offset, lib.Config.gj_tab.offset,
                   lib.Buffers.gj.offset]
    assert [lib.lanes_for(d) for d in (2, 32, 33, 100, 104, 105, 256, 416, 417, 640, 641, 1024, 1025, 2048)] == [lib.load().ptmi_lanes_for(d) for d in (2, 32, 33, 100, 104, 105, 256, 416, 417, 640, 641, 1024, 1025, 2048)]


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
