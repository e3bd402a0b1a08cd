"""TEST INFRASTRUCTURE (oracle) -- ctypes access to the plain-C restatement
(oracle/c/*.c -> oracle/_ref/liboracle_c.so).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "liboracle_c.so")


def have_coracle():
    return os.path.exists(LIB)


class COracle:
    """Backend with the interface blom_amd.hostinit / blom_amd.stepper drive."""

    def __init__(self, idm, jdm, kdm, ntr, nreg, masks):
        self.lib = C.CDLL(LIB)
        self.lib.orc_create.restype = C.c_void_p
        self.lib.orc_field.restype = C.c_void_p
        self.S = C.c_void_p(self.lib.orc_create(idm, jdm, kdm, ntr, nreg))
        self.idm, self.jdm, self.kdm, self.ntr, self.nreg = idm, jdm, kdm, ntr, nreg
        self.ni, self.nj = idm + 8, jdm + 8
        self._views = {}
        self.masks = {}
        for k in ("ip", "iu", "iv", "iq"):
            self.get(k)[0][...] = masks[k]
            self.masks[k] = self.get(k)[0]

    def get(self, name):
        if name in self._views:
            return self._views[name]
        nlev, isint = C.c_int(0), C.c_int(0)
        ptr = self.lib.orc_field(self.S, name.encode(), C.byref(nlev), C.byref(isint))
        if not ptr:
            raise KeyError(name)
        nl = nlev.value
        if name == "trc":
            nl = 2 * self.kdm * max(self.ntr, 1)
        ctype = C.c_int32 if isint.value else C.c_double
        buf = (ctype * (nl * self.nj * self.ni)).from_address(ptr)
        a = np.frombuffer(buf, dtype=np.int32 if isint.value else np.float64).reshape(nl, self.nj, self.ni)
        self._views[name] = a
        return a

    def has_stage(self, name):
        return name not in ("init_cppm", "cppm")       # cppm: the reference build itself is the oracle

    def has_field(self, name):
        try:
            self.get(name)
            return True
        except KeyError:
            return False

    def put(self, name, arr):
        v = self.get(name)
        if v is not arr:
            a = np.asarray(arr)
            v[:a.shape[0]] = a.reshape((-1,) + v.shape[1:])

    def set(self, name, v):
        if isinstance(v, str):
            rc = self.lib.orc_set_str(self.S, name.encode(), v.encode())
        elif isinstance(v, (bool, int, np.integer)):
            rc = self.lib.orc_set_int(self.S, name.encode(), C.c_int(int(v)))
            if rc:
                rc = self.lib.orc_set_real(self.S, name.encode(), C.c_double(float(v)))
        else:
            rc = self.lib.orc_set_real(self.S, name.encode(), C.c_double(float(v)))
        return rc

    def xcsum(self, name, lev=1, itype=1):
        self.lib.orc_xcsum_field.restype = C.c_double
        return self.lib.orc_xcsum_field(self.S, name.encode(), lev, itype)

    def budget_sums(self, ncall, n, nn):
        self.lib.orc_budget_sums(self.S, ncall, n, nn)

    def budget_get(self, which, ncall, n):
        self.lib.orc_budget_get.restype = C.c_double
        return self.lib.orc_budget_get(self.S, {"sdp": 0, "tdp": 1, "trdp": 2, "tkedp": 3}[which], ncall, n)

    def stage(self, name, m, n, mm, nn, k1m, k1n):
        rc = self.lib.orc_stage(self.S, name.encode(), m, n, mm, nn, k1m, k1n)
        if rc:
            raise KeyError(f"C oracle has no stage {name!r}")
