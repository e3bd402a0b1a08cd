/* blomgpu.h -- C-ABI of the MI355X-native BLOM dynamical core (libblomgpu.so).
 *
 * BLOM has no plugin/FFI layer: its operator API is the uniform Fortran stage signature
 *     subroutine <stage>(m,n,mm,nn,k1m,k1n)           phy/mod_blom_step.F90:89-253
 * acting on module-global, halo-4 arrays a(1-nbdy:idm+nbdy, 1-nbdy:jdm+nbdy, k)
 * (phy/mod_state.F90:34-86, phy/mod_xc.F90:45).  This header is what an
 * ISO_C_BINDING shim for that path binds to (INTEGRATION.md shows the shim): one entry
 * point per stage with the reference's name and argument meaning, field upload/download
 * for the module arrays, the xctilr halo update and the xccrc-style checksum.
 *
 * Conventions
 *  - all indices are the reference's 1-based Fortran values, passed by value;
 *  - a field is addressed by the reference's variable name ("dp", "utflx", "pgfxm" ...);
 *    element (i,j,k) lives at  (i+nbdy-1) + (idm+2*nbdy)*((j+nbdy-1) + (jdm+2*nbdy)*(k-1));
 *  - every function returns 0 on success, non-zero on error (the reference prints and
 *    calls xcstop/xchalt, e.g. phy/mod_advect.F90:166-172; the Fortran shim maps a
 *    non-zero status to xchalt); blomgpu_last_error() gives the text;
 *  - no host fallback exists: without a HIP device every compute entry fails.
 */
#ifndef BLOMGPU_H
#define BLOMGPU_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct blomgpu_ctx blomgpu_ctx;

/* Tile geometry, cf. module dimensions (bld/blom_dimensions:152-265) and xcspmd
 * (phy/mod_xc.F90:1332-2027): this tile owns i0+1..i0+ii, j0+1..j0+jj of the
 * itdm x jtdm global grid.  nreg as in phy/mod_bigrid.F90:81-95. */
typedef struct {
  int idm, jdm, kdm;   /* tile extents (== ii, jj) and number of layers            */
  int nbdy;            /* halo width, must be 4 (phy/mod_xc.F90:45)                 */
  int itdm, jtdm;      /* global grid size                                          */
  int i0, j0;          /* tile offsets                                              */
  int nreg;            /* 0 closed, 1 periodic-i, 2 arctic, 3 doubly periodic, 4 periodic-j */
  int ntr;             /* number of advected tracers (trc/mod_tracers.F90:225)      */
  int device;          /* HIP device ordinal                                        */
} blomgpu_dims;

int  blomgpu_create(const blomgpu_dims *dims, blomgpu_ctx **out);
int  blomgpu_destroy(blomgpu_ctx *ctx);
const char *blomgpu_last_error(const blomgpu_ctx *ctx);   /* ctx may be NULL */

/* Namelist-type options that the reference keeps as module variables set by rdlim
 * (phy/mod_rdlim.F90:137-155): baclin, batrop, delt1, dlt, lstep, nstep, pref, mdv2hi ...
 * cb, cwbdts, cwbdls, wuv1 ... wbaro, mommth, pgfmth, advmth, bmcmth, vcoord_tag ... */
int  blomgpu_set_real(blomgpu_ctx *ctx, const char *name, double v);
int  blomgpu_set_int (blomgpu_ctx *ctx, const char *name, int v);
int  blomgpu_set_str (blomgpu_ctx *ctx, const char *name, const char *v);
int  blomgpu_get_real(blomgpu_ctx *ctx, const char *name, double *v);

/* Field access.  nlev is the size of the third dimension (trc: 2*kdm*ntr). */
int  blomgpu_field_info(blomgpu_ctx *ctx, const char *name, int *nlev, int *is_int);
int  blomgpu_upload  (blomgpu_ctx *ctx, const char *name, const void *host, int nlev);
int  blomgpu_download(blomgpu_ctx *ctx, const char *name, void *host, int nlev);
/* Integer masks ip,iu,iv,iq (phy/mod_xc.F90:62-65), built by bigrid on the host. */
int  blomgpu_set_masks(blomgpu_ctx *ctx, const int *ip, const int *iu, const int *iv,
                       const int *iq);

/* xctilr(a,l1,ld,mh,nh,itype), phy/mod_xc.F90:2342 / :4222.  `lev0` (1-based) is the
 * level of `name` that the reference passes as a(1-nbdy,1-nbdy,lev0). */
int  blomgpu_xctilr(blomgpu_ctx *ctx, const char *name, int lev0, int l1, int ld,
                    int mh, int nh, int itype);

/* CRC32 of levels lev0..lev0+nlev-1 over the tile interior where the mask of the grid selected
 * by itype (1/11 p, 2/12 q, 3/13 u, 4/14 v) is 1 -- chksum/xccrc for a single tile
 * (phy/mod_checksum.F90:41-74, phy/mod_xc.F90:4164). */
int  blomgpu_crc(blomgpu_ctx *ctx, const char *name, int lev0, int nlev, int itype, unsigned *crc);

/* xccrc on a decomposed domain (phy/mod_xc.F90:2195-2322): after the halo update the reference performs there, the
 * CRCs of the 9-column strips this tile owns (those whose centre column lies in the tile), out[row 1..jdm][strip];
 * *l0 = global index of its first strip, *ns = their number.  Chaining the strips of a tile row in global order, then
 * the rows, gives the decomposition-independent checksum (blom_amd/tiles.py: chain_crc); for one tile that is
 * blomgpu_crc. */
int  blomgpu_crc_strips(blomgpu_ctx *ctx, const char *name, int lev0, int nlev, int itype, unsigned *out, int cap,
                        int *l0, int *ns);
/* Name of registered field number `index` (0-based; 1 when there is none): enumeration for hosts that move whole states. */
int  blomgpu_field_name(blomgpu_ctx *ctx, int index, char *buf, int cap);

/* Stages: same names, same argument meaning as the reference. */
int  blomgpu_init_fluxes(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n); /* phy/mod_state.F90:341   */
int  blomgpu_tmsmt1 (blomgpu_ctx *, int nn);                                             /* phy/mod_tmsmt.F90:209   */
int  blomgpu_tmsmt2 (blomgpu_ctx *, int m, int mm, int nn, int k1m);                     /* phy/mod_tmsmt.F90:281   */
int  blomgpu_initms (blomgpu_ctx *, int mm);                                             /* phy/mod_tmsmt.F90:161   */
int  blomgpu_advect (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     /* phy/mod_advect.F90:59   */
int  blomgpu_pbcor1 (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     /* phy/mod_pbcor.F90:66    */
int  blomgpu_pbcor2 (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     /* phy/mod_pbcor.F90:416   */
int  blomgpu_diffus (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     /* phy/mod_diffus.F90:41   */
int  blomgpu_pgforc (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     /* phy/mod_pgforc.F90:438  */
int  blomgpu_momtum (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     /* phy/mod_momtum.F90:215  */
int  blomgpu_convec (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);      /* phy/mod_convec.F90:43   */
/* xcsum (phy/mod_xc.F90:4116): reproducible masked sum of one level of a device field (strips of 9 points per
 * row, rows added serially); itype selects the mask as for xctilr (p-grid: the mask of the global sums, ips).
 * budget_sums (phy/mod_budget.F90:95; active with the option cnsvdi = 1): mass weighted global sums of S, T and
 * tracer 1 after call `ncall` of the step; blomgpu_budget_get(which = 0 sdp | 1 tdp | 2 trdp | 3 tkedp). */
int  blomgpu_xcsum(blomgpu_ctx *, const char *name, int lev, int itype, double *sum);
int  blomgpu_budget_sums(blomgpu_ctx *, int ncall, int n, int nn);
int  blomgpu_budget_get(blomgpu_ctx *, int which, int ncall, int n, double *value);
/* exp() as the kernels evaluate it -- the algorithm and bits of the glibc libm that the reference's compiled Fortran
 * calls (blom_amd/csrc/exp_libm.h; phy/mod_barotp.F90:183,205, phy/mod_diapfl.F90:204) -- elementwise on host arrays. */
int  blomgpu_exp(blomgpu_ctx *, int n, const double *x, double *y);
/* pow() as the kernels evaluate it: likewise the algorithm and bits of glibc's pow (blom_amd/csrc/pow_libm.h; the real powers of the
 * TKE closure and of the surface layer's stability function in difest_vertical_iso, phy/mod_difest.F90:2881-2886, :3056-3058). */
int  blomgpu_pow(blomgpu_ctx *, int n, const double *x, const double *y, double *z);
/* sin() and atan2() as the kernels evaluate them: the algorithms and bits of glibc's IBM Accurate Mathematical Library routines
 * (blom_amd/csrc/sin_libm.h, atan2_libm.h; the alignment of the flow with the topography under rhsctp, phy/mod_difest.F90:2331). */
int  blomgpu_sin(blomgpu_ctx *, int n, const double *x, double *z);
int  blomgpu_atan2(blomgpu_ctx *, int n, const double *y, const double *x, double *z);
/* phy/mod_difest.F90:735 difest_isobml, the whole routine: halo updates, interface pressure, ustar3, niw_ke_tendency (blomgpu_stage
 * "difest_isobml_pre") and the diffusivity estimates difest_common_iso (:353), difest_vertical_iso (:2629), difest_lateral_iso (:2040)
 * -> difint, difiso, difdia, difwgt (and the TKE tracers' source step).  Options: blomgpu_set_real "egc", "eggam", "eglsmn", "egmndf",
 * "egmxdf", "egidfq", "ri0", "tkepf"; blomgpu_set_int "eddf2d", "edsprs", "edanis", "redi3d", "edfsmo", "edritp_opt" (1 shear, 2 large
 * scale), "edwmth_opt" (1 smooth, 2 step), "bdmtyp", "iwdflg", "bdmldp", "rhsctp" (with blomgpu_set_real "rhiscf" and the fields
 * "betatp", "hangle": since round 6), "gls" (use_GLS, the two-equation closure: since round 6).  Inputs that
 * depend on the grid only and need the host's libm are uploaded as planes: "tdmls" (tidal mixing length scale, :2926-2927) and, with
 * bdmldp, "bdmlq" = log(2 bvf0 / max(1e-9, |coriop|)) with the option "bdml_logc" = log(2 bvf0 / cori30).
 * PARITY UNPINNED: cross-checked against the reference's real module compiled against interface-only stand-ins for CVMix. */
int  blomgpu_difest_isobml(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* the derived constants of the TKE closure (initke, phy/mod_tke.F90:133-160) as this library evaluates them, by the reference's names */
int  blomgpu_tke_const(const char *name, double *value);
int  blomgpu_updtrc (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);      /* trc/mod_tracers_update.F90:152: its idlage_step, idlage/mod_idlage.F90:57 */
int  blomgpu_sfcstr (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);      /* phy/mod_sfcstr.F90:33 (empty for channel/fuk95/noforcing) */
int  blomgpu_diapfl (blomgpu_ctx *, int n, int nn, int k1n);                             /* phy/mod_diapfl.F90:49   */
int  blomgpu_barotp (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     /* phy/mod_barotp.F90:148  */
/* phy/mod_eddtra.F90:1808 eddtra.  vcoord_type = 'isopyc_bulkml': eddtra_intdif_isopyc_bulkml / eddtra_gm_isopyc_bulkml (:153, :228).
 * The other coordinates: eddtra_ale (:1001) -- Gent-McWilliams below the mixed layer, tapered inside it, plus the mixed layer
 * restratification blomgpu_set_str "mlrmth" = "fox08" (default), "bod23" (Bodner et al. 2023, since round 5: ustar3 / wstar3 are
 * uploaded fields -- their producer for this coordinate is CVMix's KPP, which is not built) or "none"; blomgpu_set_real "ce", "tau_mlr", "tau_growing_hbl", "tau_decaying_hbl", "tau_growing_hml",
 * "tau_decaying_hml", "lfmin", "mlbl_max_ratio" (:53-94, same defaults).  Inputs by name: nslpx, nslpy (interfaces), difint, mld,
 * OBLdepth (mod_difest's boundary layer depth); state: hbl_tf, hml_tf1, hml_tf (the running means), hml_tfbnd.  Writes umfltd,
 * vmfltd, umflsm, vmflsm and the heat and salt components u/v{t,s}fl{td,sm}. */
int  blomgpu_eddtra (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* Halo updates the reference performs inside stages that are outside the hot path
 * (phy/mod_cmnfld_routines.F90:1171-1196: temp/saln halos and the kfpla halo through util1, phy/mod_difest.F90:750-772: halos + interface pressure p out to ii+3) and the
 * dp-halo/dpu/dpv tail of mxlayr (phy/mod_mxlayr.F90:1266-1310). */
/* cppm (advmth = 'cppm'): stencil tags and coefficient tables from ip, scpx, scpy; call once after the
 * grid has been uploaded.  phy/mod_cppm.F90:2504 (init_cppm, called from blom_init). */
int  blomgpu_init_cppm(blomgpu_ctx *);
int  blomgpu_halo_cmnfld2(blomgpu_ctx *, int n);
/* cmnfld2 for isopyc_bulkml (phy/mod_cmnfld_routines.F90:1158): the halo updates above plus, with eitmth = 'gm', the
 * filtered buoyancy frequency (:61-227) and the neutral slopes nslpx/nslpy, nnslpx/nnslpy (:423-652) eddtra consumes. */
int  blomgpu_cmnfld2(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* cmnfld1 for isopyc_bulkml (phy/mod_cmnfld_routines.F90:1090): cmnfld_z (:885-921), z and dz from phi, p, dp, temp, saln
   of time level m; called after tmsmt2 (phy/mod_blom_step.F90:233).  PARITY UNPINNED (the module needs netCDF). */
int  blomgpu_cmnfld1(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int  blomgpu_halo_difest (blomgpu_ctx *, int nn);
int  blomgpu_mxlayr_tail (blomgpu_ctx *, int nn, int k1n);
/* phy/mod_thermf.F90:35 thermf(m,n,mm,nn,k1m,k1n), for expcnf = 'channel' channel/mod_thermf_channel.F90:56: surface fluxes of
 * heat (surflx, sswflx, surrlx), salt (salflx, brnflx, salrlx) and tracers (trflx), the friction velocity ustar, from the forcing
 * fields swa, nsf, eva, lip, sop, rnf, rfi, ustarw and the climatologies sstclm, ricclm, sssclm; options blomgpu_set_real "trxday",
 * "srxday", "trxdpt", "srxdpt", "trxlim", "srxlim", "sref", "area", "xmi", blomgpu_set_int "l1mi".."l5mi" (mod_time's position
 * in the year), "aptflx", "apsflx", "ditflx", "disflx", "srxbal" (refused when set: not built).  Its two global sums (xcsum,
 * phy/mod_xc.F90:2071) are formed in the global domain's order on decomposed domains too: "area" is the GLOBAL ocean area, and
 * RCCL ranks need the global context of the replicated barotropic solve (blomgpu_rccl_attach_barotp_global).  PARITY UNPINNED (the module
 * imports the netCDF-bound mod_ben02): cross-checked against the real module behind a stand-in, tests/test_xcheck_thermf.py. */
int  blomgpu_thermf (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* phy/mod_mxlayr.F90:130 mxlayr(m,n,mm,nn,k1m,k1n): the bulk mixed layer of vcoord_type = 'isopyc_bulkml' (turbulent kinetic
 * energy balance, detrainment / entrainment, surface forcing, the new layer structure at the velocity points), called after
 * thermf (phy/mod_blom_step.F90:188-192).  Reads the surface fluxes surflx, surrlx, sswflx, salflx, brnflx, salrlx, trflx, the
 * friction velocity ustar and ustar3 = ustar**3, idkedt, swfc2, swal2; options blomgpu_set_real "rm0", "rm5", "niwgf", "niwbf",
 * "ce", "tau_mlr", "lfmin", "swamxd", blomgpu_set_str "mlrttp".  PARITY UNPINNED (mod_mxlayr imports netCDF-bound modules):
 * cross-checked against the real module behind stand-ins, tests/test_xcheck_mxlayr.py. */
int  blomgpu_mxlayr (blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* phy/mod_ale_regrid_remap.F90:1486 ale_regrid_remap(m,n,mm,nn,k1m,k1n): regrid the layer interfaces and remap T, S, tracers, u, v
 * (vcoord_type = 'plevel' and 'cntiso_hybrid' with regrid_method 'direct' or 'nudge'); with blomgpu_set_int
 * "ltedtp_opt" = 2 (ltedtp = 'neutral', phy/mod_diffusion.F90:123-124) also the neutral diffusion of phy/mod_ndiff.F90 between
 * regridding and remapping (option "ndiff_surface_align"; reads difiso, dpml, pu, pv; adds to utflx .. vsflx, sets utflld ..,
 * nslpx, nslpy).  PARITY: cross-checked against the real modules (mod_ale_regrid_remap behind the mod_dia stand-in,
 * mod_ndiff as it is), tests/test_xcheck_ale.py.  Options: blomgpu_set_str "vcoord_type", "ale_reconstruction_method", "ale_regrid_method",
 * "ale_tracer_limiting", "ale_velocity_limiting"; blomgpu_set_int "ale_upper_bndr_ord", "ale_lower_bndr_ord", "ale_k_range_plevel",
 * "ale_dktzu", "ale_dktzl", "ale_{density,tracer,velocity}_pc_{upper,lower}_bndr"; blomgpu_set_real "ale_dpmin_interior" [m],
 * "ale_regrid_nudge_ts", "ale_stab_fac_limit", "ale_dpvar_fac", "ale_smooth_diff_max" -- the variables of &ALE_REGRID_REMAP
 * (:1193-1201) with the reference's defaults (:69-95). */
int  blomgpu_ale_regrid_remap(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* phy/mod_ale_vdiff.F90:50 ale_vdifft, :245 ale_vdiffm: implicit vertical diffusion of T, S, tracers (with the surface fluxes and
 * their non-local transport) and of u, v.  Inputs by name: kdiff_t, kdiff_s, kvisc_m, t_{ns,sw,rs}_nonloc, s_{nb,br,rs}_nonloc
 * (kdm+1 levels), surflx, sswflx, surrlx, salflx, brnflx, salrlx, trflx (ntr planes); salt_corr, trc_corr accumulate. */
int  blomgpu_ale_vdifft(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* the halo updates of difest_lateral_hybrid (which = 0, phy/mod_difest.F90:826-831) / difest_vertical_hybrid (which = 1, :877-878) */
int  blomgpu_halo_difest_hyb(blomgpu_ctx *, int which, int k1n);
/* phy/mod_cmnfld_routines.F90:352 cmnfld_bfsqi_ale (phy/mod_blom_step.F90:199): p from dp, interface buoyancy frequency */
int  blomgpu_cmnfld_bfsqi_ale(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* phy/mod_ale_forcing.F90:45 ale_forcing: t_sw_nonloc, s_br_nonloc, buoyfl from the surface fluxes, the absorption bands swfc1,
 * swfc2, swal1, swal2 (mod_swabs) and mld (mod_cmnfld); blomgpu_set_real "swamxd", "brine_mlbase_frac" */
int  blomgpu_ale_forcing(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int  blomgpu_ale_vdiffm(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
/* 1-D module arrays: "plevel", the kdm pressure levels [g cm-1 s-2 as the model's p] of vcoord_type = 'plevel'
 * (phy/mod_vcoord.F90:99, :948-970) */
int  blomgpu_set_vector(blomgpu_ctx *, const char *name, const double *v, int nv);

/* Generic dispatcher over the entries above ("advect", "tmsmt1", ...). */
int  blomgpu_stage(blomgpu_ctx *, const char *stage, int m, int n, int mm, int nn,
                   int k1m, int k1n);

/* Device-resident time stepping: `nsteps` passes of the stage sequence of
 * phy/mod_blom_step.F90:96-253 (hot-path stages only), starting from step count
 * `nstep` (value before step_time).  Returns the new step count in *nstep.
 * Option "use_graph" = 1 (default 0: replay measures 2 % slower than plain launches on ROCm 7.2): on a single tile the
 * sequence is captured in the third step after the last change of an option or of the time step (the fourth step from rest) -- for both parities of the time levels at once -- and
 * replayed as HIP graphs from then on (steps with the stage timers on, "timing" = 1, run plain launches beside the graphs);
 * blomgpu_get_real "graph_steps" / "graph_failures" count the replays / the captures that failed. */
int  blomgpu_step(blomgpu_ctx *, int *nstep, int nsteps);

/* Tile decomposition (xcspmd/xctilr, phy/mod_xc.F90:1332-3188).  A context is one tile
 * (dims.i0/j0/itdm/jtdm).  Two transports serve blomgpu_xctilr and every halo update inside the stages:
 *  - RCCL: one process per GPU, tiles laid out along i; rank 0 creates the 128-byte id, the launcher
 *    distributes it, every rank calls blomgpu_rccl_init;
 *  - an in-process group (several tiles on one device, one host thread per tile) used by the parity tests. */
int  blomgpu_rccl_unique_id(void *id128);
int  blomgpu_rccl_init(blomgpu_ctx *, const void *id128, int rank, int nranks);          /* npx = nranks, npy = 1 */
int  blomgpu_rccl_init_2d(blomgpu_ctx *, const void *id128, int rank, int npx, int npy);  /* rank = px + npx*py */
int  blomgpu_rccl_force_ns_exchange(blomgpu_ctx *, int on);   /* test hook, see comm_rccl.hip */
/* The barotropic solve (barotp, phy/mod_barotp.F90:330-1003: 2.5 lstep substeps per step, one xctilr per substep pair in
 * the reference) replicated on every rank instead of decomposed: `global` is a second context on the same device that
 * spans the whole itdm x jtdm domain (kdm >= 3, the global masks and grid metrics uploaded); isizes[npx] / jsizes[npy]
 * are the widths / heights of the tile columns / rows.  From then on blomgpu_barotp of the tile gathers the 2-D fields
 * the solver reads (one grouped exchange), solves on `global`, and takes its window of the results. */
int  blomgpu_rccl_attach_barotp_global(blomgpu_ctx *tile, blomgpu_ctx *global, const int *isizes, const int *jsizes);
int  blomgpu_rccl_finalize(blomgpu_ctx *);
typedef struct TileGroup blomgpu_group;
int  blomgpu_group_create(int npx, int npy, blomgpu_group **out);
int  blomgpu_group_attach(blomgpu_group *, blomgpu_ctx *, int px, int py);
int  blomgpu_group_destroy(blomgpu_group *);

/* Synchronise the context's stream (all entries above are asynchronous on it). */
int  blomgpu_sync(blomgpu_ctx *);

/* Timing of the dominant kernel with HIP events on the context's stream: average
 * duration [ms] of kernel class `what` ("remap", "momtum", ...) since the last reset. */
int  blomgpu_timer_reset(blomgpu_ctx *);
int  blomgpu_timer_get(blomgpu_ctx *, const char *what, double *ms_total, int *launches);

#ifdef __cplusplus
}
#endif
#endif
