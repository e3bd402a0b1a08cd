"""CROSS-CHECK (not a pin) of the hybrid-coordinate branches of cmnfld2 and of cmnfld_bfsqi_ale against the reference's REAL
phy/mod_cmnfld_routines.F90 (cmnfld_bfsqf_ale :229-350, cmnfld_nslope_ale :654-811, cmnfld_bfsqi_ale :352-421, and cmnfld1 with cmnfld_mldl82 :933-995), compiled against
the stand-in for mod_dia in the *_xed builds (see tests/test_xcheck_cmnfld.py for why this is a cross-check).  State: the
isopycnic state after a few steps (massless layers at the bottom, sloping interfaces); the vertical coordinate switched to
'cntiso_hybrid' on both sides.  bfsqi, bfsql, bfsqf, phi, the neutral slopes and their products with the buoyancy frequency,
p, and the halos of temp and saln must agree bit for bit."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import copy_state, diff_report, fmt_report, STATE_FIELDS, GRID_FIELDS, INT_FIELDS

pytestmark = pytest.mark.gpu
OUT = ["bfsqi", "bfsql", "bfsqf", "phi", "nslpx", "nslpy", "nnslpx", "nnslpy", "temp", "saln", "p", "z", "dz", "mld", "mldl82", "dpml"]


@pytest.mark.parametrize("cfg,nsteps,eitmth", [("chan_s", 4, "gm"), ("box_s", 3, "gm"), ("fuk95", 3, "gm"), ("chan_s", 2, "intdif")])
def test_hybrid_branches_of_cmnfld_equal_the_real_module(cfg, nsteps, eitmth):
    from oracle.refblom import get_ref_backend, have_ref
    from blom_amd.gpu import BlomGpu
    lib = cfg + "_xed"
    if not have_ref(lib):
        pytest.skip(f"oracle/_ref/{lib}/libblomref.so not built")
    case = make_case(cfg, eitmth=eitmth)
    ref = get_ref_backend(lib, case.depth)
    kk = case.kdm
    gpu = BlomGpu(case.idm, case.jdm, kk, ref.ntr, ref.nreg, ref.masks)
    hostinit.init_state(gpu, case)
    assert gpu.step(0, nsteps) == nsteps
    hostinit.init_state(ref, case)
    ref.ref.set("eitmth", eitmth)
    copy_state(gpu, ref, fields=STATE_FIELDS + GRID_FIELDS + INT_FIELDS)
    for nm in OUT + ["z", "dz"]:                 # the reference's initial patterns where the stages do not write
        if ref.has_field(nm) and nm not in STATE_FIELDS:
            gpu.put(nm, ref.get(nm))
    ref.ref.set("vcoord_tag", 2)
    gpu.set("vcoord_type", "cntiso_hybrid")
    six = hostinit.step_indices(nsteps, kk)
    try:
        for st in ("cmnfld2", "cmnfld_bfsqi_ale", "cmnfld1"):
            try:
                ref.ref.stage(st, *six)
            except KeyError:
                pytest.skip("reference library built before " + st + " was added to the harness")
            gpu.stage(st, *six)
            bad = diff_report(ref, gpu, fields=OUT)
            assert not bad, st + "\n" + fmt_report(bad[:10])
        wet = ref.masks["ip"][4:-4, 4:-4] > 0
        assert (gpu.get("bfsqf")[:kk, 4:-4, 4:-4][:, wet] > 0).all()
        if eitmth == "gm":
            assert np.abs(gpu.get("nslpx")[:, 4:-4, 4:-4]).max() > 0.0
    finally:
        ref.ref.set("vcoord_tag", 1)
        gpu.close()
