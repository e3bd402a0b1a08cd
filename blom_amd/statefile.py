"""Raw binary state dump read by the Fortran driver blom_amd/fortran/blom_dyncore.F90
(format documented there).  Stands in for the reference's netCDF restart/initial-condition
files (phy/mod_restart.F90, out of scope)."""
import struct
import numpy as np

REAL_FIELDS_DEFAULT = None


def write_state(path, be, case, nsteps, fields):
    """Dump options + `fields` (names) of backend `be` so that the Fortran driver can upload them."""
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", case.idm, case.jdm, case.kdm, be.ntr, case.nreg, nsteps))
        f.write(struct.pack("<d", case.params["baclin"]))

        def entry(name, kind, nlev, payload):
            f.write(name.encode().ljust(16)[:16])
            f.write(struct.pack("<2i", kind, nlev))
            f.write(payload)
        for nm, v in case.params.items():
            if nm.endswith("0") and nm != "ri0":          # (a trailing 0 marks a parameter of the case generator; ri0 is &DIFFUSION's)
                continue
            if isinstance(v, str):
                entry(nm, 4, 0, v.encode().ljust(32)[:32])
            elif isinstance(v, (int, np.integer)):
                entry(nm, 3, 0, struct.pack("<i", int(v)))
            else:
                entry(nm, 2, 0, struct.pack("<d", float(v)))
        for m in ("ip", "iu", "iv", "iq"):
            entry(m, 1, 1, np.ascontiguousarray(be.masks[m], dtype="<i4").tobytes())
        for nm in fields:
            a = be.get(nm)
            if a.dtype.kind == "i":
                entry(nm, 1, a.shape[0], np.ascontiguousarray(a, dtype="<i4").tobytes())
            else:
                entry(nm, 0, a.shape[0], np.ascontiguousarray(a, dtype="<f8").tobytes())
