"""ctypes binding of the batched HOR3MAP C ABI (include/blomgpu_hor3map.h) -- the host-side mirror
of the reference's mod_hor3map public interface (phy/mod_hor3map.F90:268-277): the three data
structures become ReconGrid / ReconSrc / Remap objects, the procedures keep their names and
argument meaning, and every call works on all columns of a slab at once.

Arrays are numpy float64 of shape (ncol, nlev): that is the memory layout of the reference's
Fortran a(nlev, ncol).  A non-zero errstat raises Hor3mapError carrying the reference's message."""
import ctypes
import os

import numpy as np

PCM, PLM, PPM, PQM = 100, 101, 102, 103
NO_LIMITING, MONOTONIC, NON_OSCILLATORY, NON_OSCILLATORY_POSDEF = 200, 201, 203, 204
REGRID_METHOD_1, REGRID_METHOD_2 = 301, 302
P_ORD = {PCM: 0, PLM: 1, PPM: 2, PQM: 4}

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libblomgpu.so")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p


class Hor3mapError(RuntimeError):
    def __init__(self, errstat, msg):
        super().__init__(f"hor3map errstat {errstat}: {msg}")
        self.errstat = errstat


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} is missing: build it with __graft_entry__.build() "
                               "(there is no host fallback for the device library)")
        L = ctypes.CDLL(_LIB)
        ci, cd = ctypes.c_int, ctypes.c_double
        L.blomgpu_h3m_grid_create.argtypes = [ctypes.POINTER(_vp), ci, ci, ci, ci, ci, ci]
        L.blomgpu_h3m_src_create.argtypes = [_vp, ctypes.POINTER(_vp), ci, ci, ci]
        L.blomgpu_h3m_map_create.argtypes = [_vp, ctypes.POINTER(_vp), ci]
        for f in ("src_free", "map_free", "grid_free"):
            getattr(L, "blomgpu_h3m_" + f).argtypes = [_vp]
            getattr(L, "blomgpu_h3m_" + f).restype = None
        L.blomgpu_h3m_set_io.argtypes = [_vp, ci, ci]
        L.blomgpu_h3m_prepare_reconstruction.argtypes = [_vp, _vp]
        L.blomgpu_h3m_reconstruct.argtypes = [_vp, _vp, _vp]
        L.blomgpu_h3m_extract_polycoeff.argtypes = [_vp, _vp]
        L.blomgpu_h3m_regrid.argtypes = [_vp, ci, _vp, _vp, cd, ci]
        L.blomgpu_h3m_prepare_remapping.argtypes = [_vp, _vp, _vp]
        L.blomgpu_h3m_remap.argtypes = [_vp, _vp, _vp]
        L.blomgpu_h3m_reconstruct_many.argtypes = [_vp, ci, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]
        L.blomgpu_h3m_remap_many.argtypes = [ci, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp)]
        L.blomgpu_h3m_errstat.argtypes = [_vp, ctypes.POINTER(ci)]
        L.blomgpu_h3m_grid_info.argtypes = [_vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.blomgpu_h3m_sync.argtypes = [_vp]
        L.blomgpu_h3m_last_kernel_ms.argtypes = [_vp, ctypes.POINTER(ctypes.c_float)]
        L.blomgpu_h3m_errstr.argtypes = [ci]
        L.blomgpu_h3m_errstr.restype = ctypes.c_char_p
        _lib = L
    return _lib


def errstr(errstat):
    return lib().blomgpu_h3m_errstr(int(errstat)).decode()


def _check(rc, raise_on_error=True):
    if rc != 0 and raise_on_error:
        raise Hor3mapError(rc, errstr(rc))
    return rc


def _ptr(a):
    """host numpy array or raw device address (int)"""
    if isinstance(a, np.ndarray):
        assert a.dtype == np.float64 and a.flags.c_contiguous
        return a.ctypes.data
    return int(a)


class ReconGrid:
    """recon_grd_struct for ncol columns of n_src cells (mod_hor3map.F90:153)"""

    def __init__(self, ncol, n_src, method=PPM, left_bndr_ord=0, right_bndr_ord=0, device=0):
        self.ncol, self.n_src, self.method = ncol, n_src, method
        self.p_ord = P_ORD.get(method, 0)
        h = _vp()
        _check(lib().blomgpu_h3m_grid_create(ctypes.byref(h), device, ncol, n_src, method, left_bndr_ord,
                                             right_bndr_ord))
        self.h = h
        self.raise_on_error = True

    def set_io(self, device_pointers=False, check_errors=True):
        _check(lib().blomgpu_h3m_set_io(self.h, int(device_pointers), int(check_errors)))

    def prepare_reconstruction(self, x_edge_src):
        return _check(lib().blomgpu_h3m_prepare_reconstruction(self.h, _ptr(x_edge_src)), self.raise_on_error)

    def errstat(self):
        e = np.zeros(self.ncol, np.int32)
        _check(lib().blomgpu_h3m_errstat(self.h, e.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return e

    def info(self):
        n = np.zeros(self.ncol, np.int32)
        m = np.zeros(self.ncol, np.int32)
        ip = ctypes.POINTER(ctypes.c_int)
        _check(lib().blomgpu_h3m_grid_info(self.h, n.ctypes.data_as(ip), m.ctypes.data_as(ip)))
        return n, m

    def sync(self):
        _check(lib().blomgpu_h3m_sync(self.h))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        _check(lib().blomgpu_h3m_last_kernel_ms(self.h, ctypes.byref(ms)))
        return ms.value

    def free(self):
        if self.h:
            lib().blomgpu_h3m_grid_free(self.h)
            self.h = None


class ReconSrc:
    """recon_src_struct (mod_hor3map.F90:207): limiting, pc_left_bndr, pc_right_bndr"""

    def __init__(self, grid, limiting=MONOTONIC, pc_left_bndr=True, pc_right_bndr=True):
        self.grid = grid
        h = _vp()
        _check(lib().blomgpu_h3m_src_create(grid.h, ctypes.byref(h), limiting, int(pc_left_bndr),
                                            int(pc_right_bndr)))
        self.h = h

    def reconstruct(self, u_src):
        return _check(lib().blomgpu_h3m_reconstruct(self.grid.h, self.h, _ptr(u_src)), self.grid.raise_on_error)

    def extract_polycoeff(self, out=None):
        g = self.grid
        if out is None:
            out = np.zeros((g.ncol, g.n_src, g.p_ord + 1))
        rc = _check(lib().blomgpu_h3m_extract_polycoeff(self.h, _ptr(out)), g.raise_on_error)
        return out if g.raise_on_error else (out, rc)

    def regrid(self, u_edge_grd, missing_value, regrid_method=REGRID_METHOD_1, out=None, n_grd=None):
        g = self.grid
        if n_grd is None:
            n_grd = u_edge_grd.shape[1]
        if out is None:
            out = np.zeros((g.ncol, n_grd))
        rc = _check(lib().blomgpu_h3m_regrid(self.h, n_grd, _ptr(u_edge_grd), _ptr(out), missing_value,
                                             regrid_method), g.raise_on_error)
        return out if g.raise_on_error else (out, rc)


def reconstruct_many(grid, srcs, u_srcs):
    """reconstruct for several source fields of one grid in a single launch (the tracer loop)"""
    n = len(srcs)
    hs = (_vp * n)(*[s.h for s in srcs])
    ps = (_vp * n)(*[_ptr(u) for u in u_srcs])
    return _check(lib().blomgpu_h3m_reconstruct_many(grid.h, n, hs, ps), grid.raise_on_error)


def remap_many(srcs, rmap, outs=None):
    g = rmap.grid
    n = len(srcs)
    if outs is None:
        outs = [np.zeros((g.ncol, rmap.n_dst)) for _ in range(n)]
    hs = (_vp * n)(*[s.h for s in srcs])
    ps = (_vp * n)(*[_ptr(u) for u in outs])
    _check(lib().blomgpu_h3m_remap_many(n, hs, rmap.h, ps), g.raise_on_error)
    return outs


class Remap:
    """remap_struct (mod_hor3map.F90:242)"""

    def __init__(self, grid, n_dst):
        self.grid, self.n_dst = grid, n_dst
        h = _vp()
        _check(lib().blomgpu_h3m_map_create(grid.h, ctypes.byref(h), n_dst))
        self.h = h

    def prepare_remapping(self, x_edge_dst):
        return _check(lib().blomgpu_h3m_prepare_remapping(self.grid.h, self.h, _ptr(x_edge_dst)),
                      self.grid.raise_on_error)

    def remap(self, src, out=None):
        g = self.grid
        if out is None:
            out = np.zeros((g.ncol, self.n_dst))
        rc = _check(lib().blomgpu_h3m_remap(src.h, self.h, _ptr(out)), g.raise_on_error)
        return out if g.raise_on_error else (out, rc)
