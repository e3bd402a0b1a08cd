"""exp() on the device = exp() of the host libm, bit for bit.

The reference's compiled Fortran evaluates exp through glibc's libm (phy/mod_barotp.F90:183,205,
phy/mod_diapfl.F90:204).  blom_amd/csrc/exp_libm.h restates that algorithm (table of 2^(k/128), degree-5 polynomial,
fused multiply-adds where the x86-64 FMA build of glibc has them); its table is generated from first principles by
tools/gen_exp_table.py.  Checked here: the committed table is what the generator produces; the host build of the
header (tests/hostcheck/exp_hostcheck.c) and -- GPU suite -- the device through blomgpu_exp return the bits of the
host's exp() on the arguments BLOM produces and far beyond (tiny, subnormal results, overflow, inf/nan)."""
import ctypes as C
import ctypes.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libm_exp(x):
    libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.exp.restype = C.c_double
    libm.exp.argtypes = [C.c_double]
    return np.array([libm.exp(float(v)) for v in x])


def _args(n=200_000, seed=0):
    rng = np.random.default_rng(seed)
    parts = [
        rng.uniform(-40.0, 2.0, n),                    # barotp: 1 - pbu/(cwbdls onem); diapfl: -(dp..)|f| alpha0/..
        rng.uniform(-745.5, 710.0, n // 4),            # the whole finite range, subnormal results, overflow
        rng.uniform(-1.0, 1.0, n // 4) * 10.0 ** rng.integers(-20, 1, n // 4),
        np.array([0.0, -0.0, 1.0, -1.0, 1e-300, -1e-300, 2.0 ** -54, 2.0 ** -55, 511.99, 512.0, -512.0, -708.3, -708.5,
                  -744.9, -745.2, -1023.9, -1024.0, -1e5, 709.7, 709.8, 1024.0, np.inf, -np.inf, np.nan]),
    ]
    return np.concatenate(parts)


def _same_bits(a, b):
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def test_committed_table_is_what_the_generator_produces():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_exp_table
    tab = gen_exp_table.table()
    txt = open(os.path.join(ROOT, "blom_amd", "csrc", "exp_libm_table.h")).read()
    vals = [int(t.rstrip("ul,"), 16) for t in txt.split() if t.startswith("0x")]
    assert vals == tab and len(tab) == 256
    assert tab[0] == 0 and tab[1] == 0x3ff0000000000000       # 2^0 = 1 exactly, no tail


def test_host_build_returns_the_bits_of_libm_exp():
    so = os.path.join(ROOT, "tests", "hostcheck", "libexp_hostcheck.so")
    if not os.path.exists(so):
        pytest.skip("tests/hostcheck/libexp_hostcheck.so not built")
    lib = C.CDLL(so)
    x = _args()
    y = np.empty_like(x)
    lib.exp_hostcheck(C.c_int(x.size), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    want = _libm_exp(x)
    ok = _same_bits(y, want)
    assert ok.all(), [(float(v), float(a).hex(), float(b).hex()) for v, a, b in zip(x[~ok][:5], y[~ok][:5], want[~ok][:5])]


@pytest.mark.gpu
def test_device_returns_the_bits_of_libm_exp():
    from blom_amd.cases import make_case
    from blom_amd import hostinit
    from blom_amd.gpu import BlomGpu
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    x = _args(seed=1)
    y = gpu.exp(x)
    gpu.close()
    want = _libm_exp(x)
    ok = _same_bits(y, want)
    assert ok.all(), [(float(v), float(a).hex(), float(b).hex()) for v, a, b in zip(x[~ok][:5], y[~ok][:5], want[~ok][:5])]
