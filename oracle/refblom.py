"""TEST INFRASTRUCTURE (oracle) -- ctypes access to oracle/_ref/<cfg>/libblomref.so,
i.e. the reference's own Fortran stage routines (phy/mod_advect.F90 ... compiled by
oracle/Makefile) behind oracle/harness/ref_harness.F90.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Arrays are zero-copy numpy views of the reference's module storage with shape
(nlev, jdm+2*nbdy, idm+2*nbdy): Fortran a(i,j,k) == view[k-1, j+nbdy-1, i+nbdy-1].
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# cases that run on another case's reference build (same compile-time dimensions and options)
_BUILD_OF = {"fuk95_ref": "fuk95"}


def ref_lib_path(cfg):
    return os.path.join(_HERE, "_ref", _BUILD_OF.get(cfg, cfg), "libblomref.so")


def _unpack(path):
    """The channel- and tnx-sized libraries carry ~100 MB of the reference's initialised module arrays each (constant
    patterns: 1.5 MB gzipped) and the snapshot that travels to the GPU box is limited to 512 MiB, so oracle/Makefile
    leaves a libblomref.so.gz beside them and .gpurunignore keeps the unpacked files at home; unpack on first use."""
    gz = path + ".gz"
    if os.path.exists(path) or not os.path.exists(gz):
        return
    import gzip
    import shutil
    tmp = f"{path}.{os.getpid()}.tmp"
    with gzip.open(gz, "rb") as src, open(tmp, "wb") as dst:
        shutil.copyfileobj(src, dst, 1 << 24)
    os.chmod(tmp, 0o755)
    os.replace(tmp, path)


def have_ref(cfg):
    _unpack(ref_lib_path(cfg))
    return os.path.exists(ref_lib_path(cfg))


class RefBlom:
    """One instance per process and configuration (the reference keeps all state
    in Fortran module globals)."""

    def __init__(self, cfg):
        self.cfg = cfg
        # RTLD_LOCAL + distinct file => several configs can coexist in a process.
        _unpack(ref_lib_path(cfg))
        self.lib = C.CDLL(ref_lib_path(cfg), mode=os.RTLD_LOCAL | os.RTLD_NOW)
        d = (C.c_int * 8)()
        self.lib.ref_dims(d)
        self.idm, self.jdm, self.kdm, self.nbdy, self.itdm, self.jtdm, _, _ = list(d)
        self.ni = self.idm + 2 * self.nbdy
        self.nj = self.jdm + 2 * self.nbdy
        self._views = {}

    # -- scalars ---------------------------------------------------------
    def set(self, name, v):
        ierr = C.c_int(0)
        if isinstance(v, str):
            self.lib.ref_set_str(name.encode(), v.encode(), C.byref(ierr))
        elif isinstance(v, (bool, int, np.integer)):
            self.lib.ref_set_int(name.encode(), C.c_int(int(v)), C.byref(ierr))
        else:
            self.lib.ref_set_real(name.encode(), C.c_double(float(v)), C.byref(ierr))
        if ierr.value:
            raise KeyError(f"reference harness has no scalar {name!r}")

    def get_int(self, name):
        v, ierr = C.c_int(0), C.c_int(0)
        self.lib.ref_get_int(name.encode(), C.byref(v), C.byref(ierr))
        if ierr.value:
            raise KeyError(name)
        return v.value

    def get_real(self, name):
        v, ierr = C.c_double(0), C.c_int(0)
        self.lib.ref_get_real(name.encode(), C.byref(v), C.byref(ierr))
        if ierr.value:
            raise KeyError(name)
        return v.value

    # -- setup -----------------------------------------------------------
    def setup(self, depth):
        depth = np.ascontiguousarray(depth, dtype=np.float64)
        assert depth.shape == (self.nj, self.ni)
        self.lib.ref_setup(depth.ctypes.data_as(C.c_void_p))
        self._views.clear()
        self.ntr = self.get_int("ntr")
        self.nreg = self.get_int("nreg")

    # -- fields ----------------------------------------------------------
    def field(self, name):
        if name in self._views:
            return self._views[name]
        ptr, nlev, kind = C.c_void_p(), C.c_int(0), C.c_int(0)
        self.lib.ref_field(name.encode(), C.byref(ptr), C.byref(nlev), C.byref(kind))
        if not ptr.value or nlev.value <= 0:
            raise KeyError(f"reference harness has no field {name!r}")
        ctype = C.c_int32 if kind.value else C.c_double
        n = nlev.value * self.nj * self.ni
        buf = (ctype * n).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=np.int32 if kind.value else np.float64)
        a = a.reshape(nlev.value, self.nj, self.ni)
        self._views[name] = a
        return a

    def has_field(self, name):
        try:
            self.field(name)
            return True
        except KeyError:
            return False

    # -- stages ----------------------------------------------------------
    def stage(self, name, m, n, mm, nn, k1m, k1n):
        ierr = C.c_int(0)
        self.lib.ref_stage(name.encode(), m, n, mm, nn, k1m, k1n, C.byref(ierr))
        if ierr.value:
            raise KeyError(f"reference harness has no stage {name!r}")

    def xccrc(self, a, itype):
        """The reference's own checksum of a (nlev, nj, ni) float64 array (chksum/xccrc)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        crc = C.c_int(0)
        self.lib.ref_xccrc(a.ctypes.data_as(C.c_void_p), a.shape[0], itype, C.byref(crc))
        return crc.value & 0xFFFFFFFF

    def xcsum(self, a, itype=1):
        """The reference's own reproducible masked sum (xcsum) of a (nj, ni) float64 array."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        s = C.c_double(0.0)
        self.lib.ref_xcsum(a.ctypes.data_as(C.c_void_p), itype, C.byref(s))
        return s.value

    def budget_sums(self, ncall, n, nn):
        self.lib.ref_budget_sums(ncall, n, nn)

    def xctilr(self, a, l1, ld, mh, nh, itype):
        """Reference halo update on a (>=ld, nj, ni) float64 array (view), in place."""
        assert a.flags.c_contiguous and a.dtype == np.float64
        self.lib.ref_xctilr(a.ctypes.data_as(C.c_void_p), l1, ld, mh, nh, itype)


ALL_REF_FIELDS = """u v dp dpu dpv temp saln sigma uflx vflx utflx vtflx usflx vsflx p pu pv phi cau cav
ubflxs vbflxs ub vb pb pbu pbv ubflxs_p vbflxs_p pb_p pbu_p pbv_p ubcors_p vbcors_p sealv kfpla
scqx scqy scpx scpy scux scuy scvx scvy scq2 scp2 scu2 scv2 scq2i scp2i scuxi scuyi scvxi scvyi depths
corioq coriop betafp pgfx pgfy pgfx_o pgfy_o pgfxm pgfym xixp xixm xiyp xiym pgfxm_o pgfym_o xixp_o
xixm_o xiyp_o xiym_o absvor dpvor ubflx vbflx pb_mn ubflx_mn vbflx_mn pvtrop dpold dpuold dpvold
sigmar temmin difint difiso difdia difmxp difmxq difwgt umfltd vmfltd umflsm vmflsm utfltd vtfltd
utflsm vtflsm utflld vtflld usfltd vsfltd usflsm vsflsm usflld vsflld utotm vtotm utotn vtotn uflux
vflux uflux2 vflux2 uflux3 vflux3 umax vmax util1 util2 util3 util4 taux tauy ustarb trc trcold""".split()

_BACKENDS = {}


def get_ref_backend(cfg, depth, ntr=None):
    """The reference keeps all state in Fortran module globals and cannot be set up twice in
    one process; hand out one backend per configuration and restore its post-inivar state.
    ntr: give the reference's stages that many tracers to carry (ref_set_ntr in oracle/harness/ref_harness.F90: its
    tracer count is a run-time quantity, trc/mod_tracers.F90:116-126); default: the count of the build."""
    if cfg in _BACKENDS:
        be = _BACKENDS[cfg]
    else:
        be = RefBackend(cfg, depth)
        _BACKENDS[cfg] = be
    want = be.ntr_compiled if ntr is None else ntr
    if want != be.ntr:
        be.set_tracer_count(want)
    be.restore_pristine()
    return be


class RefBackend:
    """Adapter giving RefBlom the backend interface blom_amd.hostinit drives.
    Use get_ref_backend() rather than constructing this twice for one configuration."""

    def __init__(self, cfg, depth):
        self.ref = RefBlom(cfg)
        self.ref.set("expcnf", "channel")
        self.ref.setup(depth)
        self._pristine = {}
        for nm in ALL_REF_FIELDS:
            try:
                self._pristine[nm] = self.ref.field(nm).copy()
            except KeyError:
                pass
        self.kdm, self.idm, self.jdm = self.ref.kdm, self.ref.idm, self.ref.jdm
        self.ntr = self.ntr_compiled = self.ref.ntr
        self.nreg = self.ref.nreg
        self.masks = {k: self.ref.field(k)[0] for k in ("ip", "iu", "iv", "iq")}

    def restore_pristine(self):
        for nm, a in self._pristine.items():
            self.ref.field(nm)[...] = a

    def set_tracer_count(self, ntr):
        """re-allocate the reference's tracer arrays for ntr tracers; tracers beyond those it had start as copies of
        the last one's post-inivar pattern"""
        one = {nm: self._pristine[nm][:self._pristine[nm].shape[0] // self.ntr] for nm in ("trc", "trcold") if nm in self._pristine}
        self.ref.lib.ref_set_ntr(C.c_int(int(ntr)))
        self.ref._views.clear()
        self.ref.ntr = self.ref.get_int("ntr")
        assert self.ref.ntr == ntr, (self.ref.ntr, ntr)
        self.ntr = ntr
        for nm, a in one.items():
            self._pristine[nm] = np.concatenate([a] * ntr, axis=0)

    def get(self, name):
        return self.ref.field(name)          # live view: edits land in the reference

    def put(self, name, arr):
        v = self.ref.field(name)
        if v is not arr:
            v[...] = np.asarray(arr).reshape(v.shape)

    def set(self, name, v):
        try:
            self.ref.set(name, v)
        except KeyError:
            pass                             # option not read by any reference stage we call

    def has_field(self, name):
        return self.ref.has_field(name)

    def xcsum(self, a, itype=1):
        return self.ref.xcsum(a, itype)

    def budget_sums(self, ncall, n, nn):
        self.ref.budget_sums(ncall, n, nn)

    def has_stage(self, name):
        # mod_eddtra is not part of the reference build (CVMix); the cross-check builds compile it against a stand-in
        return name != "eddtra" or self.ref.cfg.endswith(("_xale", "_xaln"))

    def stage(self, name, m, n, mm, nn, k1m, k1n):
        self.ref.stage(name, m, n, mm, nn, k1m, k1n)
