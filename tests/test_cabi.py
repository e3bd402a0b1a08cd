"""The C-ABI library: it loads, exports every symbol include/blomgpu.h declares, and refuses to
compute without a HIP device (no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for hdr in ("blomgpu.h", "blomgpu_hor3map.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(blomgpu_[a-z0-9_]+)\s*\(", txt))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    from blom_amd.gpu import LIB_PATH
    if not os.path.exists(LIB_PATH):
        pytest.skip("libblomgpu.so not built (run __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_create_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from blom_amd.gpu import BlomGpu, BlomGpuError, LIB_PATH
    if not os.path.exists(LIB_PATH):
        pytest.skip("libblomgpu.so not built")
    import numpy as np
    m = {k: np.ones((16, 16), np.int32) for k in ("ip", "iu", "iv", "iq")}
    with pytest.raises(BlomGpuError, match="no HIP device"):
        BlomGpu(8, 8, 4, 0, 1, m)


def test_hor3map_grid_create_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from blom_amd import hor3map
    from blom_amd.gpu import LIB_PATH
    if not os.path.exists(LIB_PATH):
        pytest.skip("libblomgpu.so not built")
    with pytest.raises(hor3map.Hor3mapError, match="no HIP device"):
        hor3map.ReconGrid(8, 5)


def test_python_package_has_no_oracle_import():
    """The product never imports oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    for fn in os.listdir(os.path.join(ROOT, "blom_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "blom_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
