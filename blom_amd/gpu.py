"""Host-side mirror of the reference's stage interface on top of libblomgpu.so (ctypes).

`BlomGpu` offers exactly what blom_amd.hostinit / blom_amd.stepper drive (get/put/set/stage
and the integer masks), with the reference's stage names and (m,n,mm,nn,k1m,k1n) argument
meaning (phy/mod_blom_step.F90:89-253).  There is no CPU fallback: if the HIP library is
missing or no device is present, construction raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLOMGPU_LIB: another build of the same library (A/B timing of kernel variants, tools/); never a different implementation
LIB_PATH = os.environ.get("BLOMGPU_LIB") or os.path.join(_HERE, "lib", "libblomgpu.so")


class blomgpu_dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("idm", "jdm", "kdm", "nbdy", "itdm", "jtdm", "i0", "j0",
                                       "nreg", "ntr", "device")]


def load_library():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950). "
                           "blom_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.blomgpu_last_error.restype = C.c_char_p
    lib.blomgpu_last_error.argtypes = [C.c_void_p]
    return lib


class BlomGpuError(RuntimeError):
    pass


class BlomGpu:
    def __init__(self, idm, jdm, kdm, ntr, nreg, masks, device=0, itdm=None, jtdm=None, i0=0, j0=0):
        self.lib = load_library()
        self.idm, self.jdm, self.kdm, self.ntr, self.nreg = idm, jdm, kdm, ntr, nreg
        self.ni, self.nj = idm + 8, jdm + 8
        d = blomgpu_dims(idm, jdm, kdm, 4, itdm or idm, jtdm or jdm, i0, j0, nreg, ntr, device)
        self.ctx = C.c_void_p()
        rc = self.lib.blomgpu_create(C.byref(d), C.byref(self.ctx))
        if rc:
            raise BlomGpuError(self.lib.blomgpu_last_error(None).decode())
        self.masks = {k: np.ascontiguousarray(masks[k], dtype=np.int32) for k in ("ip", "iu", "iv", "iq")}
        self._chk(self.lib.blomgpu_set_masks(self.ctx, *[self.masks[k].ctypes.data_as(C.c_void_p)
                                                         for k in ("ip", "iu", "iv", "iq")]))
        self._info = {}

    def close(self):
        if self.ctx:
            self.lib.blomgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise BlomGpuError(self.lib.blomgpu_last_error(self.ctx).decode())

    # -- fields -----------------------------------------------------------------------
    def field_info(self, name):
        if name not in self._info:
            nlev, isint = C.c_int(0), C.c_int(0)
            rc = self.lib.blomgpu_field_info(self.ctx, name.encode(), C.byref(nlev), C.byref(isint))
            if rc:
                raise KeyError(name)
            self._info[name] = (nlev.value, bool(isint.value))
        return self._info[name]

    def has_stage(self, name):
        return True

    def has_field(self, name):
        try:
            self.field_info(name)
            return True
        except KeyError:
            return False

    def get(self, name):
        nlev, isint = self.field_info(name)
        a = np.empty((nlev, self.nj, self.ni), dtype=np.int32 if isint else np.float64)
        self._chk(self.lib.blomgpu_download(self.ctx, name.encode(), a.ctypes.data_as(C.c_void_p), nlev))
        return a

    def put(self, name, arr):
        nlev, isint = self.field_info(name)
        a = np.ascontiguousarray(arr, dtype=np.int32 if isint else np.float64)
        a = a.reshape((-1, self.nj, self.ni))
        n = min(nlev, a.shape[0])
        self._chk(self.lib.blomgpu_upload(self.ctx, name.encode(), a.ctypes.data_as(C.c_void_p), n))

    # -- options ----------------------------------------------------------------------
    def set(self, name, v):
        if isinstance(v, str):
            self._chk(self.lib.blomgpu_set_str(self.ctx, name.encode(), v.encode()))
        elif isinstance(v, (bool, int, np.integer)):
            rc = self.lib.blomgpu_set_int(self.ctx, name.encode(), C.c_int(int(v)))
            if rc:
                self._chk(self.lib.blomgpu_set_real(self.ctx, name.encode(), C.c_double(float(v))))
        else:
            self._chk(self.lib.blomgpu_set_real(self.ctx, name.encode(), C.c_double(float(v))))

    # -- stages -----------------------------------------------------------------------
    def stage(self, name, m, n, mm, nn, k1m, k1n):
        self._chk(self.lib.blomgpu_stage(self.ctx, name.encode(), m, n, mm, nn, k1m, k1n))

    def xctilr(self, name, lev0, l1, ld, mh, nh, itype):
        self._chk(self.lib.blomgpu_xctilr(self.ctx, name.encode(), lev0, l1, ld, mh, nh, itype))

    def set_vector(self, name, v):
        """1-D module arrays of the reference ("plevel", phy/mod_vcoord.F90:99)"""
        a = np.ascontiguousarray(v, dtype=np.float64)
        self.lib.blomgpu_set_vector.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        self._chk(self.lib.blomgpu_set_vector(self.ctx, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def step(self, nstep, nsteps=1):
        ns = C.c_int(nstep)
        self._chk(self.lib.blomgpu_step(self.ctx, C.byref(ns), nsteps))
        return ns.value

    def sync(self):
        self._chk(self.lib.blomgpu_sync(self.ctx))

    def crc(self, name, lev0, nlev, itype=1):
        v = C.c_uint(0)
        self._chk(self.lib.blomgpu_crc(self.ctx, name.encode(), lev0, nlev, itype, C.byref(v)))
        return v.value

    def crc_strips(self, name, lev0, nlev, itype=1):
        """This tile's share of the decomposition-independent checksum (xccrc, phy/mod_xc.F90:2195-2322): (l0, strips)
        with strips[row, s] the CRC of its s-th own 9-column strip of the global rows; chain with tiles.chain_crc."""
        cap = self.jdm * ((self.idm + 8) // 9 + 2)
        out = np.zeros(cap, dtype=np.uint32)
        l0, ns = C.c_int(0), C.c_int(0)
        self._chk(self.lib.blomgpu_crc_strips(self.ctx, name.encode(), lev0, nlev, itype, out.ctypes.data_as(C.c_void_p),
                                              cap, C.byref(l0), C.byref(ns)))
        return l0.value, out[:self.jdm * ns.value].reshape(self.jdm, ns.value).copy()

    def field_names(self):
        """every registered field (reals, then integers): the arrays a whole state consists of"""
        names, buf, k = [], C.create_string_buffer(64), 0
        while self.lib.blomgpu_field_name(self.ctx, k, buf, 64) == 0:
            names.append(buf.value.decode())
            k += 1
        return names

    def xcsum(self, name, lev=1, itype=1):
        """xcsum (phy/mod_xc.F90:4116) of one level of a device field; p-grid mask: ips."""
        v = C.c_double(0.0)
        self._chk(self.lib.blomgpu_xcsum(self.ctx, name.encode(), lev, itype, C.byref(v)))
        return v.value

    def exp(self, x):
        """exp() as the kernels evaluate it (blom_amd/csrc/exp_libm.h), elementwise."""
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        y = np.empty_like(x)
        self._chk(self.lib.blomgpu_exp(self.ctx, x.size, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)))
        return y

    def get_real(self, name):
        v = C.c_double(0.0)
        self._chk(self.lib.blomgpu_get_real(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def pow(self, x, y):
        """pow() as the kernels evaluate it (blom_amd/csrc/pow_libm.h), elementwise."""
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        z = np.empty_like(x)
        self._chk(self.lib.blomgpu_pow(self.ctx, x.size, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p)))
        return z

    def sin(self, x):
        """sin() as the kernels evaluate it (blom_amd/csrc/sin_libm.h), elementwise."""
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        z = np.empty_like(x)
        self.lib.blomgpu_sin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._chk(self.lib.blomgpu_sin(self.ctx, x.size, x.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p)))
        return z

    def atan2(self, y, x):
        """atan2() as the kernels evaluate it (blom_amd/csrc/atan2_libm.h), elementwise."""
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        z = np.empty_like(x)
        self.lib.blomgpu_atan2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.blomgpu_atan2(self.ctx, x.size, y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p)))
        return z

    def tke_const(self, name):
        """a derived constant of the TKE closure (phy/mod_tke.F90:133-160) as this library evaluates it"""
        v = C.c_double(0.0)
        if self.lib.blomgpu_tke_const(name.encode(), C.byref(v)):
            raise KeyError(name)
        return v.value

    def budget_sums(self, ncall, n, nn):
        """budget_sums (phy/mod_budget.F90:95); does nothing unless the option cnsvdi is set."""
        self._chk(self.lib.blomgpu_budget_sums(self.ctx, ncall, n, nn))

    def budget_get(self, which, ncall, n):
        v = C.c_double(0.0)
        self._chk(self.lib.blomgpu_budget_get(self.ctx, {"sdp": 0, "tdp": 1, "trdp": 2, "tkedp": 3}[which], ncall, n, C.byref(v)))
        return v.value

    # -- tile decomposition ---------------------------------------------------------------
    def rccl_init(self, id128, rank, nranks):
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._chk(self.lib.blomgpu_rccl_init(self.ctx, buf, rank, nranks))

    def rccl_init_2d(self, id128, rank, npx, npy):
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._chk(self.lib.blomgpu_rccl_init_2d(self.ctx, buf, rank, npx, npy))

    def rccl_attach_barotp_global(self, glob, isizes, jsizes):
        """every rank solves the whole 2-D barotropic domain on `glob`, a BlomGpu of the global domain (blomgpu.h)"""
        a = (C.c_int * len(isizes))(*[int(x) for x in isizes])
        b = (C.c_int * len(jsizes))(*[int(x) for x in jsizes])
        self._chk(self.lib.blomgpu_rccl_attach_barotp_global(self.ctx, glob.ctx, a, b))
        self._bt_global = glob               # keep it alive as long as the tile

    def rccl_force_ns_exchange(self, on=True):
        self._chk(self.lib.blomgpu_rccl_force_ns_exchange(self.ctx, int(on)))

    def rccl_finalize(self):
        self.lib.blomgpu_rccl_finalize(self.ctx)

    def timer_reset(self):
        self._chk(self.lib.blomgpu_timer_reset(self.ctx))

    def timer_get(self, what):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.lib.blomgpu_timer_get(self.ctx, what.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def rccl_unique_id():
    lib = load_library()
    buf = (C.c_char * 128)()
    if lib.blomgpu_rccl_unique_id(buf):
        raise BlomGpuError("ncclGetUniqueId failed")
    return bytes(buf)


class TileGroup:
    """Several tiles of one domain on ONE device in one process (one host thread per tile): the
    in-process halo transport used to check decomposition independence (tests)."""

    def __init__(self, npx, npy):
        self.lib = load_library()
        self.h = C.c_void_p()
        self.lib.blomgpu_group_create(npx, npy, C.byref(self.h))
        self.npx, self.npy = npx, npy

    def attach(self, gpu, px, py):
        gpu._chk(self.lib.blomgpu_group_attach(self.h, gpu.ctx, px, py))

    def destroy(self):
        self.lib.blomgpu_group_destroy(self.h)
