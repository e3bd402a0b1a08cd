"""Stage sequencing of one baroclinic step of the dynamical core.

Mirrors the order of phy/mod_blom_step.F90:96-253 for the isopyc_bulkml branch,
restricted to the stages of the hot path and the ones sitting between them
(SURVEY.md 8a/8f).  convec (phy/mod_blom_step.F90:174, between momtum and diapfl) is part of it.  Stages whose
reference modules need netCDF/CVMix (cmnfld, difest, mxlayr, thermf) are not part of the sequence: diffusivities and isopycnal slopes stay frozen.  eddtra is
part of it; the reference build used as oracle cannot contain it (mod_eddtra needs mod_difest ->
CVMix), so a backend without it skips it -- with zero slopes nslpx/nslpy, which is what the
reference-pinned cases use, eddtra's result is exactly the zero fluxes that backend keeps.
"""
from .hostinit import step_indices

DYNCORE_STAGES = ("init_fluxes", "tmsmt1", "halo_cmnfld2", "halo_difest", "eddtra", "advect", "pbcor1",
                  "diffus", "pgforc", "momtum", "convec", "diapfl", "mxlayr_tail", "updtrc", "barotp", "pbcor2",
                  "tmsmt2")
OPTIONAL_STAGES = ("eddtra",)
# halo_cmnfld2 / halo_difest : the xctilr calls of phy/mod_cmnfld_routines.F90:1171-1172 and
#   phy/mod_difest.F90:750-755 (the stages themselves are out of scope, their halo updates
#   are not: advect and momtum read those halos).
# mxlayr_tail : phy/mod_mxlayr.F90:1266-1310, halo of dp at the new level + dpu/dpv.


# Config 2 "full blom_step" as far as it is built: cmnfld2 on the live state, the part of difest_isobml in front
# of the diffusivity estimates (halos, interface pressure, ustar3, niw_ke_tendency: "difest_isobml_pre"), thermf and mxlayr in
# place of the mxlayr_tail pseudo-stage, cmnfld1 at the end -- the order of phy/mod_blom_step.F90:126-233 for isopyc_bulkml.
# Still frozen: the diffusivities themselves (difest_common/vertical/lateral_iso need CVMix at module level).
FULL_STAGES = tuple({"halo_cmnfld2": "cmnfld2", "halo_difest": "difest_isobml_pre"}.get(s, s) for s in DYNCORE_STAGES
                    if s != "mxlayr_tail")
FULL_STAGES = FULL_STAGES[:FULL_STAGES.index("diapfl") + 1] + ("thermf", "mxlayr") + FULL_STAGES[FULL_STAGES.index("diapfl") + 1:] + ("cmnfld1",)

# ... and with the diffusivity estimates of difest_isobml live (difest_common_iso, difest_vertical_iso, difest_lateral_iso; the device
# option difest_live): the whole routine in place of its front part
FULL_STAGES_LIVE = tuple("difest_isobml" if s == "difest_isobml_pre" else s for s in FULL_STAGES)

STAGES_FROZEN_EDDY_FLUXES = tuple(s for s in DYNCORE_STAGES if s != "eddtra")    # umfltd.. stay as uploaded

# The step of the other vertical coordinates (vcoord_type = 'cntiso_hybrid' or 'plevel'), phy/mod_blom_step.F90:126-233, as far
# as it is built: ale_regrid_remap in front, the ALE column physics (cmnfld_bfsqi_ale, ale_forcing, ale_vdifft, ale_vdiffm) in
# place of convec / diapfl / mxlayr, cmnfld1 at the end (the mixed layer depth ale_forcing reads).  Left out, as their modules
# need CVMix: difest_lateral_hybrid, difest_vertical_hybrid (the diffusivities, the non-local fractions and the boundary layer
# depth stay as uploaded); thermf (the surface fluxes stay as uploaded).  eddtra runs its ALE form (eddtra_ale).
# halo_difest_hyb / halo_difest_vert: the xctilr calls of the two routines that are left out (phy/mod_difest.F90:826-831, :877-878).
HYBRID_STAGES = ("init_fluxes", "tmsmt1", "ale_regrid_remap", "cmnfld2", "halo_difest_hyb", "eddtra", "advect", "pbcor1", "diffus", "pgforc",
                 "momtum", "cmnfld_bfsqi_ale", "ale_forcing", "halo_difest_vert", "ale_vdifft", "ale_vdiffm", "updtrc", "barotp",
                 "pbcor2", "tmsmt2", "cmnfld1")


def dyncore_step(be, nstep, baclin, stages=DYNCORE_STAGES, hook=None):
    """Advance backend `be` from step count nstep to nstep+1.  `hook(stage, sextuple)` is
    called before each stage (tests use it to snapshot stage inputs)."""
    six = step_indices(nstep, be.kdm)
    be.set("nstep", nstep + 1)               # step_time, phy/mod_blom_step.F90:99
    for st in stages:
        if st in OPTIONAL_STAGES and not getattr(be, "has_stage", lambda s: True)(st):
            continue
        if hook is not None:
            hook(st, six)
        be.stage(st, *six)
    be.set("delt1", baclin + baclin)         # phy/mod_blom_step.F90:300
    return nstep + 1
