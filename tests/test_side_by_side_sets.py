"""The pairs of kernel chains blomgpu_step runs side by side on its two streams (option phys_dag) must share no array that
one side writes: checked here on the SOURCES (tools/kernel_rw_sets.py: fields, integer fields and work-space slots a kernel binds to a
mutable pointer or stores through are its writes, every other mention a read), so that an edit which makes a kernel of one chain touch an
array of the other fails on the CPU, before a race on the GPU has to be caught by bytes (tests/test_gpu_variants.py::
test_physics_stages_side_by_side_on_the_second_stream).  The halo updates that run inside a window are listed with the fields they write."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_rw_sets import rw_sets  # noqa: E402

CSRC = os.path.join(ROOT, "blom_amd", "csrc")
CMN, DFI, DFE, DIA, THF, MXL, SIM = ("stage_cmnfld.hip", "stage_difest_iso.hip", "stage_difest.hip", "stage_diapfl.hip", "stage_thermf.hip",
                                     "stage_mxlayr.hip", "stage_simple.hip")

# (name, kernels on the second stream, kernels beside them on the context's stream, fields the halo updates inside the window write there)
WINDOWS = [
    ("cmnfld2 beside difest's front part and vertical chain (bit 1; the pressure scan that writes p goes in front of the fork)",
     [(CMN, "k_cmn_bfsqf"), (CMN, "k_cmn_phi"), (CMN, "k_cmn_nslope")],
     [(DFE, "k_difest_ustar3"), (DFE, "k_niw_uv"), (DFE, "k_niw_idkedt"), (DFI, "k_dfi_kmax_kfil"), (DFI, "k_dfi_kfil_util"), (DFI, "k_dfi_uv2"),
      (DFI, "k_dfi_common"), (DFI, "k_dfi_vert_a"), (DFI, "k_dfi_vert_b"), (DFI, "k_dfi_vert_c")],
     {"F_u", "F_v", "F_ubflxs_p", "F_vbflxs_p", "F_pbu", "F_pbv", "F_util1"}),
    ("difest's lateral part beside its vertical chain (bit 1)",
     [(DFI, "k_dfi_falign"), (DFI, "k_dfi_lateral")],
     [(DFI, "k_dfi_vert_a"), (DFI, "k_dfi_vert_b"), (DFI, "k_dfi_vert_c")], set()),
    ("diapfl's momentum mixing beside thermf_channel and mxlayr's first kernels (bit 2)",
     [(DIA, "k_diapfl_momentum"), (DIA, "k_diapfl_dpudpv")],
     [(THF, "k_thermf_channel_flux"), (THF, "k_thermf_channel_corr"), (MXL, "k_mxl_bg2_sig"), (MXL, "k_mxl_bg2_grad"), (MXL, "k_mxl_bg2_sum")],
     {"F_util1"}),
    ("mxlayr's copy-back clamp beside the pressures and thicknesses at the velocity points (bit 8; the halo update of dp writes halo points "
     "only without the arctic patch, the kernel reads interior points)",
     [(MXL, "k_mxl_clamp")], [(SIM, "k_pscan"), (SIM, "k_dpudpv")], set()),
]


@pytest.mark.parametrize("name,side,main,halo", WINDOWS, ids=[w[0].split(" (")[0] for w in WINDOWS])
def test_chains_side_by_side_share_no_written_array(name, side, main, halo):
    def sets(chain):
        r, w = set(), set()
        for f, k in chain:
            rr, ww = rw_sets(os.path.join(CSRC, f), k)
            assert rr | ww, (f, k, "no array found: the parser lost the kernel")
            r |= rr | ww
            w |= ww
        return r, w

    rs, ws = sets(side)
    rm, wm = sets(main)
    wm |= halo
    rm |= halo
    masks = {"I_ip", "I_iu", "I_iv", "I_iq"}                       # constant after initialisation
    clash = ((ws & rm) | (wm & rs)) - masks
    assert not clash, f"{name}: {sorted(clash)} written on one side and touched on the other"


def test_barotp_touches_no_tracer():
    """updtrc's ideal-age step runs beside barotp's first kernels (bit 4), the clamp's tracers until pbcor2 (bit 8)."""
    for f in ("stage_barotp.hip", "stage_barotp_pair.hip"):
        src = open(os.path.join(CSRC, f)).read()
        for fld in ("F_trc", "F_saln", "F_salt_corr", "F_trc_corr"):
            assert fld not in src, (f, fld)


def test_the_parser_sees_a_known_conflict():
    """k_dfi_common reads what k_dfi_uv2 writes (work slots 0 and 1, msku / mskv): the two are NOT side by side, and the check must say so."""
    _, w = rw_sets(os.path.join(CSRC, DFI), "k_dfi_uv2")
    r, _ = rw_sets(os.path.join(CSRC, DFI), "k_dfi_common")
    assert {"wk:0", "wk:1", "I_msku", "I_mskv"} <= (w & r)
    # a store through a device helper's pointer argument counts: the pressure scan writes p, which cmnfld2's kernels read -- it is why
    # st_cmnfld2 launches difest's scan in front of the fork
    _, wp = rw_sets(os.path.join(CSRC, SIM), "k_pscan")
    rb, _ = rw_sets(os.path.join(CSRC, CMN), "k_cmn_bfsqf")
    assert "F_p" in wp and "F_p" in rb
    # cmnfld2's work slot lies behind difest's nine (it was slot 3 = difest's W_BVF until the two ran side by side)
    _, wb = rw_sets(os.path.join(CSRC, CMN), "k_cmn_bfsqf")
    assert "wk:9" in wb and not any(x in wb for x in ("wk:%d" % i for i in range(9)))
