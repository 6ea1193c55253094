"""The table-driven draws of the MH path (oracle/ptmcmc_oracle.c orc_unit_log / orc_unit_sincos*, kernels: ptmi_device.h):
tables identical for kernels and oracle and reproducible from tools/make_draw_tables.py, accuracy against libm, sign and
symmetry properties the sampler relies on."""
import ctypes as C
import math
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def _numbers(path):
    txt = open(path).read()
    return [float.fromhex(t) for t in re.findall(r"-?0x[0-9a-f]\.[0-9a-f]+p[+-]\d+", txt)]


def test_tables_identical_and_reproducible():
    dev = _numbers(os.path.join(ROOT, "ptmcmcsampler_amd", "csrc", "ptmi_tables.h"))
    ora = _numbers(os.path.join(ROOT, "oracle", "orc_tables.h"))
    assert len(dev) == 128 and dev == ora
    mp = pytest.importorskip("mpmath")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_draw_tables as mk
    lt, st = mk.tables()
    assert [v for pair in lt for v in pair] + [v for pair in st for v in pair] == dev
    # sincos table: exact mirror symmetry, unit norm
    c, s = np.array(dev[64::2]), np.array(dev[65::2])
    assert np.array_equal(c[16:], c[:16][::-1]) and np.array_equal(s[16:], -s[:16][::-1])   # theta -> 2 pi - theta
    assert np.array_equal(c[8:16], -c[:8][::-1]) and np.array_equal(s[8:16], s[:8][::-1])   # theta -> pi - theta
    assert np.abs(c * c + s * s - 1).max() < 3e-16


def _ulps(a, b):
    return abs(a - b) / math.ulp(b) if b != 0 else abs(a)


def test_unit_log_accuracy_and_sign():
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 120
    L = orc.lib()
    rs = np.random.RandomState(11)
    ws = [int(x) for x in rs.randint(0, 2 ** 63, size=20000, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)]
    ws += [0, 2 ** 64 - 1, 2 ** 11 - 1, 2 ** 11, (2 ** 53 - 1) << 11, (2 ** 53 - 2) << 11, (2 ** 52) << 11, ((2 ** 52) - 1) << 11]
    ws += [((2 ** 53 - 1 - k) << 11) for k in range(1, 2000)]        # u just below 1: the result must not turn positive
    ws += [(k << 11) for k in range(0, 2000)]                        # the far tail
    ws += [int(2 ** 64 * f) - 1 for f in np.linspace(0.69, 0.70, 500)] + [int(2 ** 63 * f) for f in np.linspace(1.385, 1.395, 500)]
    worst = 0.0
    for w in ws:
        n = (w >> 11) + 1
        got = L.orc_unit_log(w)
        want = float(mp.log(mp.mpf(n) / 2 ** 53))
        assert got <= 0.0 and (got < 0.0 or n == 2 ** 53)
        worst = max(worst, _ulps(got, want))
    assert worst <= 2.0, worst


def test_unit_sincos_accuracy_and_symmetry():
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 120
    L = orc.lib()
    rs = np.random.RandomState(12)
    sn, cs = C.c_double(), C.c_double()
    worst = 0.0
    ws = [0, 2 ** 64 - 1, 2 ** 59, 2 ** 59 - 1] + [int(x) for x in rs.randint(0, 2 ** 63, size=20000, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)]
    for k, w in enumerate(ws):
        L.orc_unit_sincos64(w, C.byref(sn), C.byref(cs))
        j, m = w >> 59, (w >> 7) & (2 ** 52 - 1)
        if k < 3000:
            ang = 2 * mp.pi * (mp.mpf(j) + mp.mpf(m) / 2 ** 52) / 32
            worst = max(worst, abs(cs.value - float(mp.cos(ang))), abs(sn.value - float(mp.sin(ang))))
        # half a turn further: both exactly negated (the proposal distribution is exactly symmetric)
        s2, c2 = C.c_double(), C.c_double()
        L.orc_unit_sincos64((w + 2 ** 63) % 2 ** 64, C.byref(s2), C.byref(c2))
        assert s2.value == -sn.value and c2.value == -cs.value
    assert worst < 4e-16, worst
    for h in [int(x) for x in rs.randint(0, 2 ** 32, size=5000, dtype=np.int64)] + [0, 2 ** 32 - 1, 2 ** 27, 2 ** 27 - 1]:
        L.orc_unit_sincos32(h, C.byref(sn), C.byref(cs))
        ang = 2 * mp.pi * mp.mpf(h) / 2 ** 32
        assert abs(cs.value - float(mp.cos(ang))) < 4e-16 and abs(sn.value - float(mp.sin(ang))) < 4e-16


def test_unit_normals_moments():
    L = orc.lib()
    rs = np.random.RandomState(13)
    w = rs.randint(0, 2 ** 63, size=(100000, 2), dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    zc, zs = C.c_double(), C.c_double()
    z = np.empty((len(w), 2))
    for i, (a, b) in enumerate(w):
        L.orc_unit_normals(int(a), int(b), C.byref(zc), C.byref(zs))
        z[i] = zc.value, zs.value
    z32 = np.array([L.orc_unit_normal32(int(a), int(b) & 0xFFFFFFFF) for a, b in w[:50000]])
    for v in (z[:, 0], z[:, 1], z32):
        n = len(v)
        assert abs(v.mean()) < 4.5 / math.sqrt(n) and abs(v.var() - 1) < 4.5 * math.sqrt(2.0 / n)
        assert abs((v ** 4).mean() - 3) < 0.15
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.015
