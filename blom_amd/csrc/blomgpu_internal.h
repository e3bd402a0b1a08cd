// Internal declarations of libblomgpu.so (HIP, gfx950).  Not part of the C-ABI.
//
// Data layout in HBM: every BLOM module array keeps the reference's layout
//   a(1-nbdy:idm+nbdy, 1-nbdy:jdm+nbdy, k), i fastest           (phy/mod_state.F90:34-86)
// so that a level is one contiguous (ni x nj) plane, rows are unit-stride in i (coalesced
// across a wavefront) and host<->device transfers are plain memcpys of Fortran storage.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>
#include "exp_libm.h"

#define NBDY 4

// every vector-memory operation of this wave has completed (inline asm: invisible to the compiler pass that would
// otherwise drop the wait behind a release fence, MI355X_MICROARCH.md "Compiler hazard").  tests/hostemu compiles
// these sources for the host to check kernel logic on the CPU; there the wait is empty.
#ifdef BLOM_HOSTEMU
#define WAIT_VMCNT0() ((void)0)
#else
#define WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// ---- field table -------------------------------------------------------------------
// X(name, levels) with levels in terms of K = kdm and NT = ntr.
#define BLOM_REAL_FIELDS(X)                                                              \
  /* mod_state */                                                                        \
  X(u, 2 * K) X(v, 2 * K) X(dp, 2 * K) X(dpu, 2 * K) X(dpv, 2 * K) X(temp, 2 * K)        \
  X(saln, 2 * K) X(sigma, 2 * K) X(uflx, 2 * K) X(vflx, 2 * K) X(utflx, 2 * K)           \
  X(vtflx, 2 * K) X(usflx, 2 * K) X(vsflx, 2 * K)                                        \
  X(p, K + 1) X(pu, K + 1) X(pv, K + 1) X(phi, K + 1) X(cau, K) X(cav, K)                \
  X(ubflxs, 3) X(vbflxs, 3) X(ub, 2) X(vb, 2) X(pb, 2) X(pbu, 2) X(pbv, 2)               \
  X(ubflxs_p, 2) X(vbflxs_p, 2) X(pb_p, 1) X(pbu_p, 1) X(pbv_p, 1) X(ubcors_p, 1)        \
  X(vbcors_p, 1) X(sealv, 1)                                                             \
  /* mod_grid */                                                                         \
  X(scqx, 1) X(scqy, 1) X(scpx, 1) X(scpy, 1) X(scux, 1) X(scuy, 1) X(scvx, 1)           \
  X(scvy, 1) X(scq2, 1) X(scp2, 1) X(scu2, 1) X(scv2, 1) X(scq2i, 1) X(scp2i, 1)         \
  X(scuxi, 1) X(scuyi, 1) X(scvxi, 1) X(scvyi, 1) X(corioq, 1) X(coriop, 1)              \
  X(betafp, 1) X(depths, 1)                                                              \
  /* mod_pgforc */                                                                       \
  X(pgfx, 2 * K) X(pgfy, 2 * K) X(pgfx_o, K) X(pgfy_o, K) X(pgfxm, 2) X(pgfym, 2)        \
  X(xixp, 2) X(xixm, 2) X(xiyp, 2) X(xiym, 2) X(pgfxm_o, 1) X(pgfym_o, 1)                \
  X(xixp_o, 1) X(xixm_o, 1) X(xiyp_o, 1) X(xiym_o, 1)                                    \
  /* mod_momtum */                                                                       \
  X(absvor, 2 * K) X(dpvor, 2 * K)                                                       \
  /* mod_barotp (module arrays + the implicitly SAVEd locals, mod_barotp.F90:155-167) */ \
  X(ubflx, 2) X(vbflx, 2) X(pb_mn, 2) X(ubflx_mn, 2) X(vbflx_mn, 2) X(pvtrop, 2)         \
  X(pvtrop_o, 1) X(pb_t, 2) X(ubflx_t, 2) X(vbflx_t, 2) X(pb_t2, 2) X(ubflx_t2, 2) X(vbflx_t2, 2) X(umaxb, 1) X(uminb, 1)          \
  X(vmaxb, 1) X(vminb, 1) X(uglue, 1) X(vglue, 1) X(ubflxs_t, 1) X(vbflxs_t, 1)          \
  X(ubcors_t, 1) X(vbcors_t, 1)                                                          \
  /* mod_tmsmt */                                                                        \
  X(dpold, 2 * K) X(dpuold, K) X(dpvold, K) X(told, K) X(sold, K)                        \
  /* mod_vcoord / mod_diffusion */                                                       \
  X(sigmar, K) X(sigint, K) X(temmin, K) X(difint, K) X(difiso, K) X(difdia, K) X(difmxp, 1) X(difmxq, 1)          \
  X(difwgt, 1) X(umfltd, 2 * K) X(vmfltd, 2 * K) X(umflsm, 2 * K) X(vmflsm, 2 * K)       \
  X(utfltd, 2 * K) X(vtfltd, 2 * K) X(utflsm, 2 * K) X(vtflsm, 2 * K) X(utflld, 2 * K)   \
  X(vtflld, 2 * K) X(usfltd, 2 * K) X(vsfltd, 2 * K) X(usflsm, 2 * K) X(vsflsm, 2 * K)   \
  X(usflld, 2 * K) X(vsflld, 2 * K)                                                      \
  /* mod_utility / mod_forcing */                                                        \
  X(utotm, 1) X(vtotm, 1) X(utotn, 1) X(vtotn, 1) X(uflux, 1) X(vflux, 1) X(uflux2, 1)   \
  X(vflux2, 1) X(uflux3, 1) X(vflux3, 1) X(umax, 1) X(vmax, 1) X(util1, 1) X(util2, 1)   \
  X(util3, 1) X(util4, 1) X(taux, 1) X(tauy, 1) X(ustarb, 1)                             \
  /* ale_vdifft / ale_vdiffm (mod_ale_vdiff.F90): vertical diffusivities and non-local transport fractions at the layer     \
     interfaces (mod_diffusion.F90:131-139, mod_forcing.F90:181-191), surface fluxes (mod_forcing.F90:159-164), the salt and  \
     tracer corrections (:177-179), the tracers' surface fluxes (mod_tracers.F90:60; here one plane per tracer) */            \
  X(kvisc_m, K + 1) X(kdiff_t, K + 1) X(kdiff_s, K + 1) X(t_ns_nonloc, K + 1) X(s_nb_nonloc, K + 1) X(t_sw_nonloc, K + 1)    \
  X(t_rs_nonloc, K + 1) X(s_br_nonloc, K + 1) X(s_rs_nonloc, K + 1) X(surflx, 1) X(sswflx, 1) X(surrlx, 1) X(salflx, 1)     \
  X(brnflx, 1) X(salrlx, 1) X(salt_corr, 1) X(trc_corr, NT) X(trflx, NT)                                                    \
  /* ale_forcing (mod_ale_forcing.F90): the two-band shortwave absorption of mod_swabs, the mixed layer depth of mod_cmnfld    \
     [m], the interface buoyancy flux (mod_forcing.F90:183) */                                                                \
  /* momtum, the other vertical coordinates: the fractions of the wind stress that pass the interfaces (mod_diffusion.F90:139-142) */ \
  X(mu_nonloc, K + 1) X(mv_nonloc, K + 1)                                                                                     \
  X(swfc1, 1) X(swfc2, 1) X(swal1, 1) X(swal2, 1) X(mld, 1) X(mldl82, 1) X(dpml, 1) X(buoyfl, K + 1) X(hbl_tf, 1) X(hml_tf1, 1) X(hml_tf, 1) X(hml_tfbnd, 1) X(OBLdepth, 1)                                                 \
  /* thermf_channel (channel/mod_thermf_channel.F90): the forcing fields and the climatologies of mod_forcing it reads / sets */   \
  X(swa, 1) X(nsf, 1) X(hmltfz, 1) X(lip, 1) X(sop, 1) X(eva, 1) X(rnf, 1) X(rfi, 1) X(fmltfz, 1) X(sfl, 1) X(ustarw, 1)           \
  X(sstclm, 12) X(ricclm, 12) X(sssclm, 12)                                                                                      \
  /* niw_ke_tendency (mod_niw.F90:52-65): mixed layer velocities of the last steps and their running-mean reservoirs */          \
  X(uml, 4) X(vml, 4) X(umlres, 2) X(vmlres, 2)                                                                                   \
  /* mxlayr (mod_mxlayr.F90:70-91 diagnostics; mod_forcing ustar, ustar3; mod_niw idkedt) */                                  \
  X(ustar, 1) X(ustar3, 1) X(wstar3, 1) X(wpup_tf, 1) X(idkedt, 1) X(mtkeus, 1) X(mtkeni, 1) X(mtkebf, 1) X(mtkers, 1) X(mtkepe, 1) X(mtkeke, 1) X(pbrnda, 1)   \
  /* mod_tracers: trc(i,j,2*kdm,ntr), trcold(i,j,kdm,ntr) */                             \
  X(trc, 2 * K * NT) X(trcold, K * NT)                                                   \
  /* mod_diapfl SAVEd arrays (mod_diapfl.F90:59) */                                      \
  X(fpug, K) X(fplg, K) X(nslpx, K) X(nslpy, K) X(nnslpx, K) X(nnslpy, K) X(bfsqi, K + 1) X(bfsql, K) X(bfsqf, K + 1)                                                                 \
  /* mod_cmnfld: depth of the layer interfaces and layer thickness [m] (cmnfld1) */    \
  X(z, K + 1) X(dz, K)                                                                   \
  /* mod_cppm: thickness edge values and the coefficient tables of init_cppm (mod_cppm.F90:79-89);     \
     the j-tables in (i,j) order (the reference's "_perm" layout, :2511-2518) */                        \
  X(hel_3d, K) X(her_3d, K) X(hevc1i, 1) X(hevc2i, 1) X(hevc3i, 1) X(hevc4i, 1) X(ssci, 1) X(scci, 1)    \
  X(d2mi, 1) X(tmc0i, 12) X(tmcli, 12) X(tmcri, 12) X(hevc1j, 1) X(hevc2j, 1) X(hevc3j, 1) X(hevc4j, 1)   \
  X(sscj, 1) X(sccj, 1) X(d2mj, 1) X(tmc0j, 12) X(tmclj, 12) X(tmcrj, 12)                                                                  \
  /* difest_isobml's diffusivity estimates (stage_difest_iso.hip): mod_grid's latitude, topographic beta and grid angles; mod_difest's \
     rig, du2l, drhol; mod_tke's Prod, Buoy, Shear2, L_scale; mod_tidaldissip's twedon, mod_seaice's ficem; and two planes the host     \
     fills with its own libm at initialisation: the tidal mixing length scale (a tanh of the latitude, mod_difest.F90:2926-2927) and      \
     log(2 bvf0 / max(1e-9, |f|)) of the latitude dependent background mixing (:2747-2750) */                                              \
  X(plat, 1) X(betatp, 1) X(cosang, 1) X(sinang, 1) X(hangle, 1) X(twedon, 1) X(ficem, 1) X(tdmls, 1) X(bdmlq, 1)                          \
  X(rig, K + 1) X(du2l, K) X(drhol, K) X(Prod, K) X(Buoy, K) X(Shear2, K) X(L_scale, K)                                                    \
  /* (K+1)-level work fields (phip of pgforc_geopotential, ...) */                       \
  X(wkp0, K + 1) X(wkp1, K + 1)

#define BLOM_INT_FIELDS(X) X(ip, 1) X(iu, 1) X(iv, 1) X(iq, 1) X(kfpla, 2) X(kming, 1) X(cppm_sti, 1) X(cppm_stj, 1) X(mpack, 1) \
  /* mod_difest: kmax, kfil, msku, mskv (mod_difest.F90:97-100) */ X(dfe_kmax, 1) X(dfe_kfil, 1) X(msku, K) X(mskv, K)

enum FieldId {
#define X(name, lev) F_##name,
  BLOM_REAL_FIELDS(X)
#undef X
      NF_REAL
};
enum IFieldId {
#define X(name, lev) I_##name,
  BLOM_INT_FIELDS(X)
#undef X
      NF_INT
};

// ---- scalar options (namelist-type module variables of the reference) ----------------
struct Params {
  // mod_time
  double baclin, batrop, delt1, dlt;
  int lstep, nstep;
  int nday_in_year, itriag;   // mod_time; index of the ideal age tracer (trc/mod_tracers.F90:100), < 1: none
  // the reference's TKE build options (phy/mod_ifdefs.F90:16-35) as run-time switches: itrtke >= 1 <=> use_TKE;
  // itrtke, itrgls: tracer indices (trc/mod_tracers.F90:87-91); tkeadv, tkeidf, gls <=> use_TKEADV, use_TKEIDF, use_GLS
  int itrtke, itrgls, tkeadv, tkeidf, gls;
  // mod_eos (inieos, phy/mod_eos.F90:105-129)
  double pref;
  double ap11, ap12, ap13, ap14, ap15, ap16, ap21, ap22, ap23, ap24, ap25, ap26;
  // mod_momtum
  double mdv2hi, mdv2lo, mdv4hi, mdv4lo, mdc2hi, mdc2lo, vsc2hi, vsc2lo, vsc4hi, vsc4lo;
  double cbar, cb;
  // mod_barotp
  double cwbdts, cwbdls;
  // mod_tmsmt
  double wuv1, wuv2, wts1, wts2, wbaro;
  // mod_diffusion (diapfl)
  double bdmc1, bdmc2, iwdfac, nubmin;
  int bdmtyp, iwdflg, bdmldp;
  // option codes
  int mommth;      // 0 enscon, 1 enecon, 2 enedis          (phy/mod_momtum.F90:723-765)
  int pgfmth;      // 0 geopotential, 1 dynamic enthalpy    (phy/mod_pgforc.F90:525-527)
  int advmth;      // 0 remap, 1 cppm                       (phy/mod_advect.F90:96,155)
  int bmcmth;      // 0 uc, 1 dluc                          (phy/mod_pbcor.F90:99-105)
  int eitmth;      // 1 intdif, 2 gm                        (phy/mod_diffusion.F90:112-113)
  int vcoord_tag;  // 1 isopyc_bulkml                       (phy/mod_vcoord.F90)
  int ltedtp_opt;  // 1 layer, 2 neutral                    (phy/mod_diffusion.F90)
  double vland;    // halo fill value for closed boundaries (phy/mod_xc.F90:104)
  int allwet;      // momtum's viscous march: take the all-wet form of a step where the masks allow (A/B option, default 1)
};
// tracers (1-based nt) left out of layer diffusion (phy/mod_diffus.F90:64-66) and of advection (phy/mod_remap.F90:314-316)
__host__ __device__ inline bool trc_skip_dif(const Params &P, int nt) { return P.itrtke >= 1 && !P.tkeidf && (nt == P.itrtke || nt == P.itrgls); }
__host__ __device__ inline bool trc_skip_adv(const Params &P, int nt) { return P.itrtke >= 1 && !P.tkeadv && (nt == P.itrtke || nt == P.itrgls); }
constexpr double TKE_MIN = 7.6e-8, GLS_PSI_MIN = 1.e-14;      // phy/mod_tke.F90:61-62

// ---- what a kernel sees ----------------------------------------------------------------
// The field pointers a kernel takes out of its DevView are pointers into HBM -- but a pointer LOADED from memory has the generic
// address space, and hipcc (ROCm 7.2) then emits flat_load / flat_store for everything reached through it: a flat access counts on
// vmcnt AND lgkmcnt and may complete out of order, so the compiler waits for `vmcnt(0) lgkmcnt(0)` wherever a loaded value is used --
// for the wave's YOUNGEST access, whatever was requested ahead of it (tools/isa_waits.py shows it; until round 6 every kernel of this
// library was built that way: 706 flat against 26 global loads in stage_pgforc.o alone).  The tables below hand the pointers out in
// the global address space (device code only; the conversion to a plain `double *` at the use is implicit), which turns every such
// access into global_load / global_store with counted waits (`s_waitcnt vmcnt(n)`).  Same arithmetic, same results.
#if defined(__HIP_DEVICE_COMPILE__)
#define BLOM_GAS __attribute__((address_space(1)))
#else
#define BLOM_GAS
#endif
template <class T, int N> struct PtrTable {
  T *p_[N];
  __host__ __device__ inline T BLOM_GAS *operator[](int i) const { return (T BLOM_GAS *)p_[i]; }
  __host__ inline T *&operator[](int i) { return p_[i]; }
};
template <class T> __host__ __device__ inline T BLOM_GAS *global_ptr(T *p) { return (T BLOM_GAS *)p; }
// Local pointer variables of a kernel that are used WITHOUT an offset added first (`const double *p = V.f[F_p]`) must keep the
// address space in their type: the round trip global -> generic right after the table's cast is folded away by the optimiser
// and the accesses are flat again (tools/flat_census.py lists the kernels that still have some).
typedef const double BLOM_GAS *gcd_t;
typedef double BLOM_GAS *gd_t;
typedef const int BLOM_GAS *gci_t;
typedef int BLOM_GAS *gi_t;

struct DevView {
  int ii, jj, kk;        // tile extents (== idm, jdm, kdm)
  int ni, nj;            // padded plane: idm+2*nbdy, jdm+2*nbdy
  int nplane;            // ni*nj
  int itdm, jtdm, i0, j0, nreg, ntr;
  Params P;
  PtrTable<double, NF_REAL> f;
  PtrTable<int, NF_INT> m;
  // work space standing in for the reference's stage-local temporaries:
  //   wk   : nwk fields of kk levels each   (field w, level k at wk + (w*kk + k)*nplane)
  //   wk2d : NWK2D single planes
  double *wk;
  double *wk2d;
  int nwk;
};
#define NWK2D 48
#define WK(V, w) (global_ptr((V).wk) + (size_t)(w) * (V).kk * (V).nplane)
#define WK2(V, w) (global_ptr((V).wk2d) + (size_t)(w) * (V).nplane)

// index of Fortran (i,j) inside a plane; level stride is V.nplane
#define IDX(V, i, j) (((i) + NBDY - 1) + (V).ni * ((j) + NBDY - 1))

// Fortran MAX/MIN as amdflang lowers them (fcmp ogt/olt + select): on ties -- in
// particular (+0,-0) -- the SECOND operand is returned.
__host__ __device__ inline double fmax2(double a, double b) { return a > b ? a : b; }
__host__ __device__ inline double fmin2(double a, double b) { return a < b ? a : b; }
__host__ __device__ inline double fmax3(double a, double b, double c) { return fmax2(fmax2(a, b), c); }
__host__ __device__ inline double fmin3(double a, double b, double c) { return fmin2(fmin2(a, b), c); }

// per-wavefront phase timestamps (debug builds with -DBLOM_KPROF only: `make kprof`; blomgpu_dbg_kprof, tools/kprof_waves.py): wave w
// writes words 8 w .. 8 w + 7; KPROF_PASS(id) hands the buffer to the kernel the option kprof_sel names and nullptr to the others
#ifdef BLOM_KPROF
#define KPROF_ARGS , long long *kprof, int kprof_words
#define KPROF_PASS(id) , (c->kprof_sel == (id) ? c->kprof : nullptr), c->kprof_words
#define KPROF_MARK(wave, slot) do { if (kprof && 8 * (wave) + (slot) < kprof_words) kprof[8 * (wave) + (slot)] = wall_clock64(); } while (0)
#define KPROF_ADD(wave, slot, v) do { if (kprof && 8 * (wave) + (slot) < kprof_words) atomicAdd((unsigned long long *)&kprof[8 * (wave) + (slot)], (unsigned long long)(v)); } while (0)
// one count per trip of the WAVE through this point (the lowest active lane counts)
#define KPROF_TRIP(wave, slot) do { if (kprof && 8 * (wave) + (slot) < kprof_words && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) kprof[8 * (wave) + (slot)] += 1; } while (0)
#else
#define KPROF_ARGS
#define KPROF_PASS(id)
#define KPROF_MARK(wave, slot) do { } while (0)
#define KPROF_ADD(wave, slot, v) do { } while (0)
#define KPROF_TRIP(wave, slot) do { } while (0)
#endif

// ---- context ---------------------------------------------------------------------------
struct KTimer {
  double ms = 0.0;
  int launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// ---- tile decomposition / halo transport (mod_xc's xcspmd + xctilr, phy/mod_xc.F90:1332-3188) ------
struct blomgpu_ctx;
struct TileGroup;            // in-process transport: several tiles (contexts) on one device, one host thread each
struct RcclComm;             // one process per GPU, neighbour ncclSend/ncclRecv over xGMI
struct BtGlobal;             // RCCL tiles: the barotropic solve replicated on every rank (comm_rccl.hip)
struct Tiling {
  int npx = 1, npy = 1;      // uniform tile grid (mproc x nproc)
  int px = 0, py = 0;        // this tile
  TileGroup *group = nullptr;
  RcclComm *rccl = nullptr;
  bool multi() const { return npx * npy > 1 || rccl != nullptr; }
};

struct blomgpu_ctx {
  DevView h;                 // host copy (pointers are device pointers)
  DevView *d = nullptr;      // device copy handed to kernels
  bool dirty = true;         // host copy changed since last upload
  int device = 0;
  hipStream_t stream = nullptr;
  // second stream for halo exchanges that overlap with compute (RCCL tiles: barotp), its fork/join events,
  // and the stream the RCCL transport enqueues on when set (otherwise `stream`)
  hipStream_t xstream = nullptr, halo_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int barotp_rimbuf = 1;    // RCCL tiles: barotp's pair kernel reads its E/W rim from the receive buffers
  int barotp_arctic_persist = 1; // arctic patch, single tile: the odd+even pairs of a phase in the persistent launch (stage_barotp.hip)
  int barotp_arctic_fused = 1;   // decomposed arctic domain: fused pair kernel with the single tile's exchange schedule
  int barotp_overlap = 0;   // measured slower (see stage_barotp_pair.hip: bt_overlap_usable)
  int nlev_real[NF_REAL];
  int nlev_int[NF_INT];
  std::unordered_map<std::string, int> real_ids, int_ids;
  std::unordered_map<std::string, KTimer> timers;
  bool timing = false;
  Tiling tiling;
  bool cppm_ready = false;   // init_cppm has run (tables on the device)
  int cppm_compat = 1;       // 1 full, 2 partial            (phy/mod_cppm.F90:55-58)
  int cppm_limiting = 2;     // 1 monotonic, 2 non_oscillatory
  // device error words: [0] diapfl, [1] eddtra, [2] barotp abort.  Stage entries read theirs back at once;
  // blomgpu_step defers the read-back to the end of the step (one host sync per step instead of three)
  int *err_dev = nullptr;
  bool defer_checks = false;
  // blomgpu_step replays the stage sequence of a step as a HIP graph (one per parity of the time levels), captured from
  // the stream once the lazily allocated buffers exist; any option / parameter / mask change drops the graphs
  int halo_overlap = 1;          // RCCL tiles: the halo exchange in front of remap runs on xstream while the inner tiles compute
                                 // (bit-identical; ON by default since round 4, the arctic patch included -- in self-send on one GPU it
                                 // costs 0.2 ms, 9.78 against 9.59 ms per step; what it buys over xGMI has not been measured)
  int cmnfld1 = 0;               // blomgpu_step ends with cmnfld1 (z, dz of the new state; consumed by diagnostics and difest only)
  int use_graph = 0;             // measured slower than plain launches on ROCm 7.2 (channel 8.25 vs 7.90 ms, tnx2v1s 5.44 vs 5.12): off by default
  hipGraphExec_t step_graph[2] = {nullptr, nullptr};
  int steps_done = 0;
  BtGlobal *bt_global = nullptr; // RCCL tiles: barotp gathers its 2-D inputs and solves the global domain on a second context
  bool stream_borrowed = false;  // this context runs on another context's stream (the global barotropic context of a tile)
  bool in_sequence = false;      // blomgpu_step is running its stage sequence: a stage may hand data to the next one through the work space
  bool pbcor2_handed_over = false;   // likewise pbcor2 (level m) for tmsmt2
  bool pbcor2_dp_in_wk = false;      // ... and its column pass (the rescaling of dp) is left to tmsmt2 as well
  int tmsmt_fold = 1;                // option: 0 = pbcor2 always runs its column pass
  bool pbcor1_handed_over = false;   // pbcor1 left S, T, tracers of the new level in the work space (slots N_S, N_T, N_TR): diffus starts there
  int steps_warm = 0;            // plain steps since the last option change (graph capture waits for 2: both time levels' buffers exist)
  long long graph_steps = 0;     // steps replayed as graphs (blomgpu_get_real "graph_steps")
  int graph_failures = 0;        // captures that began and failed (tried again after two more plain steps, three times at most)
  int pgf_uv_pair = 0;           // k_pgf_uv, A/B option: 1 = the u- and v-column wavefronts of 64 points in one workgroup with XCD-contiguous
                                 // numbering: fetch 1.44 -> 0.92 GB per launch, but 0.388 against 0.369 ms for the stage (same box, two runs
                                 // each): the kernel waits on its k-serial chain, not on bytes.  Off.
  int scan_reassoc = 0;          // TOLERANCE-MODE EXPERIMENT (off): k_pscan with k on the lanes and a log-step shuffle prefix sum -- NOT bit-identical to the reference (stage_simple.hip)
  hipStream_t side3 = nullptr;   // the viscous march's land-side pairs beside its all-wet pairs (mom_aw_split = 2)
  hipEvent_t ev_side3[2] = {nullptr, nullptr};
  int mom_aw_split = 0;          // the viscous march's all-wet (chunk, strip) pairs on a launch of their own with the mask-free kernel (round 6): 1 = one after the other, 2 = side by side on a third stream; measured, off
  int mom_aw_key = -1, mom_aw_n[2] = {0, 0};
  int *mom_aw_dev = nullptr;     // the two launch lists, device memory
  int mom_force_aw = 0;          // TIMING EXPERIMENT ONLY: k_mom_visc_march with the mask-free body alone (wrong next to land)
  int convec_nsingle = 2;        // k_convec_velocity: moves of a level walked singly before chunks of old layers are requested (A/B; 1000 = the kernel of rounds 1-5)
  int cmn_nslope_nb = 4;         // k_cmn_nslope: interfaces in flight in the interior sweep (A/B: 2, 3, 4)
  int pgf_reuse = 0;             // k_pgf_uv: skip the equation of state where a level repeats the previous level's inputs (wave-uniform; bit-identical)
  int pgf_uv_ring = 7;           // k_pgf_uv_ring (round 6: every load of the level loop statically countable): 1 = separate u / v workgroups, 2 = paired + XCD-contiguous
  int pgf_copy_fused = 1;        // pgforc: the pgfx_o/pgfy_o copy rides along in k_pgf_uv
  int eddtra_frozen = 0;         // blomgpu_step leaves eddtra out: umfltd, vmfltd, umflsm, vmflsm stay as uploaded
  int check_period = 8;          // steps between read-backs of the sticky stage error words in blomgpu_step
  bool csdiag = false;           // mod_checksum's switch: error words read back every step
  bool bt_restart = true;        // the next persistent barotp launch zeroes the completion counts and starts at epoch 0
  unsigned bt_epoch = 0;          // completion count every tile has reached after the launches so far
  // blomgpu_step with more steps to come: tmsmt2 also writes what the next step's tmsmt1 would copy (stage_simple.hip)
  // option: inside blomgpu_step k_remap_tile also does k_remap_update's work (stage_remap_tile.hip).  Off: measured equal
  // (channel 6.41-6.45 against 6.43-6.45 ms per step, remap 0.910 against 0.906 ms) to slower (tnx2v1s 3.71 against 3.66 ms) -- the
  // overlapping tiles cost the tile kernel what the update kernel took, though 1.1 GB less crosses HBM
  int remap_fold = 0;
  int remap_nfirst = 4;      // more than 4 advected tracers: how many of them ride with dp, T, S in k_remap_tile's first pass (0..4; 4 measured fastest)
  bool remap_handed_over = false;   // ... and has left dp, T, S, tracers of the new level in the work space for pbcor1
  // ale_regrid_remap (stage_ale.hip): the options of &ALE_REGRID_REMAP with the reference's defaults (mod_ale_regrid_remap.F90:69-95),
  // as hor3map codes (include/blomgpu_hor3map.h); the engine's structures; the pressure levels of vcoord_type = 'plevel'
  int ale_method = 102, ale_upper_bndr_ord = 6, ale_lower_bndr_ord = 4, ale_tracer_limiting = 203, ale_velocity_limiting = 203;
  bool ale_tracer_pc_upper = true, ale_tracer_pc_lower = false, ale_velocity_pc_upper = true, ale_velocity_pc_lower = false;
  bool ale_density_pc_upper = false, ale_density_pc_lower = false;
  int ale_regrid_method = 2, ale_k_range_plevel = 1;      // 'nudge' is the reference's default
  double ale_regrid_nudge_ts = 86400., ale_stab_fac_limit = .75, ale_dpvar_fac = .75, ale_smooth_diff_max = 50000.;   // :80-85
  int ale_dktzu = 4, ale_dktzl = 2;
  double ale_dpmin_interior = .1 * 9806.;                 // [m] in the namelist, times onem (:1352-1353)
  // mod_mxlayr's namelist variables (cime_config/namelist_definition_blom.xml: rm0 = 1.2, rm5 = 0, mlrttp = 'constant'), mod_niw's
  double rm0 = 1.2, rm5 = 0., niwgf = 0., niwbf = .35, niwlf = .5;
  std::string mlrttp = "constant";
  // thermf (mod_forcing's namelist variables, phy/mod_forcing.F90:43-62, :84; mod_grid's area; mod_time's position in the year,
  // phy/mod_time.F90: xmi, l1mi..l5mi; mod_ben02's ntda)
  double trxday = 0., srxday = 0., trxdpt = 1., srxdpt = 1., trxlim = 1.5, srxlim = .5, sref = 34.65, area = 0., xmi = 0.;
  int lmi[5] = {11, 12, 1, 2, 3};
  bool aptflx = false, apsflx = false, ditflx = false, disflx = false, srxbal = false;
  int ntda = 0;
  bool full_physics = false;  // blomgpu_step runs difest_isobml_pre, thermf, mxlayr (+ cmnfld2, cmnfld1): stepper.py FULL_STAGES
  double swamxd = 200., brine_mlbase_frac = 0.;           // phy/mod_swabs.F90:183 (default); phy/mod_forcing.F90:63 (namelist)
  void *ale = nullptr;
  double *ale_plevel = nullptr;
  // eddtra_ale (stage_eddtra_ale.hip): the options of &DIFFUSION that phy/mod_eddtra.F90 owns, with its defaults (:53-94)
  int mlrmth = 1;                                         // 0 none, 1 fox08, 2 bod23 (hybrid coordinate only)
  double eddtra_ce = .06, tau_mlr = 86400., tau_growing_hbl = 300., tau_decaying_hbl = 86400., tau_growing_hml = 3600.,
         tau_decaying_hml = 259200., lfmin = 5.e3, mlbl_max_ratio = 3., eddtra_cl = .25, mstar = .5, nstar = .066, wpup_min = 1.e-3;
  bool fluxes_lean = false;      // in sequence: init_fluxes zeroed only the ring the storing tile kernel of remap leaves out; cleared by that kernel's launch
  bool fluxes_zeroed = false;    // in sequence: init_fluxes has run and remap has not yet (its u-faces then add to zero)
  bool tmsmt1_ahead = false, tmsmt1_done_ahead = false;
  int tmsmt_ahead = 1;           // option: 0 = every step launches its own tmsmt1
  unsigned *bt_flags = nullptr;   // abort word + per-tile completion counts of the persistent barotp kernel
  int num_cus = 0;
  int bt4_blocks_per_cu[2] = {-1, -1};          // the same for the temporally blocked kernel (k_bt_steps4)
  int bt_blocks_per_cu[4] = {-1, -1, -1, -1};   // occupancy query results for the persistent barotp kernel's shapes (-1: not asked yet)
  int barotp_block = 1;      // 1: four substeps per hand-off where the persistent form runs (k_bt_steps4: no arctic patch, one process); 2: the same
                             // kernel with one launch per four substeps (any number of tiles; the host emulation); 0: off
  int barotp_persist = 1;    // 1: one launch per barotropic phase where all tiles are resident (stage_barotp_pair.hip)
  long long *bt_prof = nullptr;   // debug: phase timestamps of k_bt_pair
  long long *kprof = nullptr;     // debug (builds with -DBLOM_KPROF): per-wavefront phase timestamps of a column kernel, 8 words a wave (blomgpu_dbg_kprof)
  int kprof_words = 0;
  int kprof_sel = 1;              // which marked kernel writes: 1 k_pgf_uv*, 2 k_diapfl_column3, 3 k_mxl_column, 4 k_convec_column, 5 k_eddtra_gm, 6 k_mom_column_from, 7 k_diapfl_momentum, 9 k_remap_tile, 10 k_pbc_tile (these two: a word set per WORKGROUP of level 10)
  int diffus_shfl = 0;       // A/B: west neighbours of diffus' flux kernel through wavefront shuffles
  int ndiff_rec_per_face = 0;    // neutral diffusion: records per face (0: 6 kk, the bound; stage_ale.hip)
  int ndiff_surface_align = 1;   // phy/mod_diffusion.F90:84 (the namelist default of cime_config is .true.)
  int live_slopes = 0;       // blomgpu_step: 1 = cmnfld2 computes nslpx/nslpy every step (stage_cmnfld.hip); 0 = they stay as uploaded
  int momtum_order = 0;      // A/B: 0 chunk-major work order of the fused kernels, 1 layer-major
  int momtum_lds_pad = 0;    // experiment: extra bytes of LDS per workgroup of the one-wavefront marches (lowers the occupancy)
  int momtum_bs = 0;         // lanes per workgroup of the fused kernels (0: 64, one wavefront)
  int momtum_chunks_a = 0, momtum_chunks_b = 0;   // j-chunks per layer of the two fused kernels (0: one round of workgroups)
  int diapfl_du = 8;         // levels whose loads k_diapfl_column3 keeps in flight (2, 4, 8)
  int barotp_tile = 0;       // tile shape of the pair kernel, 100*TI + TJ (3216, 3208, 1608); 0: chosen from the tile count
  int barotp_fused = 1;      // 1: LDS-tiled substep pairs (stage_barotp_pair.hip), 0: one kernel per equation
  double *arc_strip = nullptr;                 // arctic patch, tiles of one process in strips mode: this tile's strip
  size_t arc_cap = 0;
  int arctic_strips = 0;                       // tiles of one process: arctic fold through packed strips (test of the RCCL path's kernels)
  double *xcsum_dev = nullptr;                 // sums that stay on the device (thermf), 8 slots
  double *xcsum_buf = nullptr;                 // [0] the sum, [1..jj] the row sums of xcsum
  int cnsvdi = 0;                              // mod_budget: conservation diagnostics on/off
  double budget[4][7][2] = {};                 // sdp, tdp, trdp, tkedp (ncall, n)
  std::string err;
  std::string expcnf = "channel";   // experiment configuration (mod_config): selects the forcing branches
  // ---- stage overlap inside blomgpu_step (round 5) ----------------------------------------------------------------
  // The column kernels of the step (one thread per column, 1.6 wavefronts per SIMD on the channel) wait on their k-serial
  // chains and leave the issue slots and the memory system idle; kernels of ANOTHER stage that do not depend on them run
  // beside them on a second stream.  What makes stages independent is that every stage of the reference rebuilds the
  // interface pressures p, pu, pv for itself: momtum's kernels get private copies (alternative device views of the
  // context: the same fields, with p / pu / pv / the work space pointing at buffers of momtum's own), so its viscous chain
  // -- which reads nothing the stages between difest and momtum write -- can start right after difest's halo updates, and
  // convec's column kernel -- which writes p from dp(kn) -- can run beside momtum's Coriolis kernel, which reads p from dp(km).
  int overlap = 1;               // option: 0 = every stage on the context's stream, in order
  // inside blomgpu_step (remap): init_fluxes only zeroes the faces remap does not store, the tile kernel's mass / heat / salt fluxes
  // go to uflx .. vsflx alone and k_remap_update reads them there (12 F of stores less per step); 0: the work planes of rounds 1-4
  int lean_fluxes = 1;
  // difest_isobml's diffusivity estimates (stage_difest_iso.hip): the variables of &DIFFUSION that phy/mod_difest.F90 reads
  // (phy/mod_diffusion.F90:45-110), with the values of the reference's own tests/fuk95/limits as defaults
  double egc = 0., eggam = 200., eglsmn = 4000., egmndf = 0., egmxdf = 1500., egidfq = 1., rhiscf = 0., ri0 = 1.2, tkepf = 0.;
  double bdml_logc = 0.;         // log(2 bvf0 / cori30) of the host's libm (set with the plane bdmlq when bdmldp is on)
  int eddf2d = 0, edsprs = 1, edanis = 0, redi3d = 0, rhsctp = 0, edfsmo = 0, edritp_opt = 2, edwmth_opt = 1;
  int difest_live = 0;           // blomgpu_step: 1 = difest_isobml estimates difint, difiso, difdia, difwgt every step (full_physics)
  hipStream_t side = nullptr;    // the second stream
  hipEvent_t ev_side[10] = {};
  DevView hv[4];                 // host copies of the alternative views (hv[0] unused: the main view is h)
  DevView *dv[4] = {nullptr, nullptr, nullptr, nullptr};
  double *wk_mom = nullptr, *p_alt = nullptr, *pu_alt = nullptr, *pv_alt = nullptr;
  bool mom_early_done = false;   // in sequence: momtum's viscous chain of this step is already running on `side`
  bool convec_col_ahead = false; // in sequence: convec's column kernel of this step is already running on `side`
  // phys_dag: in sequence, cmnfld2's three column kernels run on `side` beside difest's common part and its vertical chain, and
  // difest's lateral part (falign, lateral) follows them there (stage_cmnfld.hip: st_cmnfld2, stage_difest_iso.hip: st_difest_isobml)
  int phys_dag = 7;              // (bit 8 measured: 0.02 ms on the channel, inside the noise -- off)
  int mom_early_at = 0;          // momtum's viscous chain forks behind: 0 difest (halo_difest), 1 eddtra, 2 advect, 3 pbcor1, 4 diffus
  bool cmn_on_side = false;      // in sequence: cmnfld2's kernels of this step are on `side`, nothing has waited for them yet
  // the same option's bits 2 and 4, small launches beside long ones: diapfl's momentum mixing beside thermf and mxlayr's first kernels (2),
  // updtrc's ideal-age step beside barotp's first kernels (4), mxlayr's copy-back clamp beside the rest of mxlayr (8; `updtrc_on_side` says that
  // something on the second stream has to be waited for in front of pbcor2).  (pgforc's p / dpu / dpv beside diffus' tile kernel: measured, dropped --
  // a kernel that fills the chip leaves a second queue only its tail, whatever the queue's priority)
  bool diapfl_mom_on_side = false, updtrc_on_side = false;
  bool pscan_done_ahead = false; // in sequence: st_cmnfld2 has launched the pressure scan of difest_isobml's front part (blomgpu_halo_difest)
};
// alternative device views (see blomgpu_ctx::overlap): MOM_A = momtum's work space + its pu, pv (k_mom_pupv, the viscous
// march); MOM_B = momtum's work space + its p (k_mom_pscan, k_mom_drag, the Coriolis march); MOM_C = momtum's work space
// with the model's own p, pu, pv (the vertical pass, which leaves pu, pv of the new time level in the module arrays)
enum { VIEW_MAIN = 0, VIEW_MOM_A, VIEW_MOM_B, VIEW_MOM_C, NVIEW };
static inline const DevView *ctx_view(const blomgpu_ctx *c, int which) { return which == VIEW_MAIN ? c->d : c->dv[which]; }
int ctx_side_fork(blomgpu_ctx *c, int ev);     // `side` waits for what is on the context's stream now
int ctx_side_done(blomgpu_ctx *c, int ev);     // marks the end of a section on `side` ...
int ctx_side_join(blomgpu_ctx *c, int ev);     // ... and the context's stream waits for it
bool ctx_overlap_on(const blomgpu_ctx *c);

int  ctx_fail(blomgpu_ctx *c, const std::string &msg);
int  ctx_pack_masks(blomgpu_ctx *c);     // mpack = ip | iu << 1 | iv << 2 | iq << 3, after any upload of a mask
void ctx_sync_view(blomgpu_ctx *c);      // uploads h -> d if dirty
void ctx_drop_graphs(blomgpu_ctx *c);    // forget the captured step graphs (an option, parameter or mask changed)
#define HIPCHK(c, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return ctx_fail((c), std::string(#call) + ": " + hipGetErrorString(e_));            \
  } while (0)

// timing scope around a kernel class (HIP events on the context stream)
struct TimeScope {
  blomgpu_ctx *c;
  hipEvent_t a = nullptr, b = nullptr;
  const char *what;
  hipStream_t st = nullptr;      // the stream the bracketed launches go to (default: the model's)
  TimeScope(blomgpu_ctx *c_, const char *w, hipStream_t s = nullptr);
  ~TimeScope();
};

// ---- stage implementations (one translation unit per stage) ---------------------------
int st_init_fluxes(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_tmsmt1(blomgpu_ctx *, int nn);
int st_tmsmt2(blomgpu_ctx *, int m, int mm, int nn, int k1m);
int st_initms(blomgpu_ctx *, int mm);
int st_advect(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_pbcor1(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_pbcor2(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_diffus(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_pgforc(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_momtum(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_momtum_fused_layers(blomgpu_ctx *, int m, int n, int mm, int nn, int part);   // part: 0 all, 1 the viscous march, 2 the rest
int st_momtum_early(blomgpu_ctx *, int m, int n, int mm, int nn);   // in sequence: the viscous chain of this step's momtum on the second stream
int st_convec_column_ahead(blomgpu_ctx *, int n, int nn);         // in sequence: convec's column kernel on the second stream
int st_diapfl(blomgpu_ctx *, int n, int nn, int k1n);
int st_convec(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_updtrc(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_xcsum(blomgpu_ctx *, const double *a, int itype, double *sum);
int st_xcsum_dev(blomgpu_ctx *, const double *a, int itype, int slot, double **sums_dev);   // the sum stays on the device: sums_dev[slot]
int st_thermf(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);   // stage_thermf.hip
int st_niw_ke_tendency(blomgpu_ctx *, int m, int mm);                           // stage_difest.hip
int st_difest_isobml_pre(blomgpu_ctx *, int m, int n, int mm, int nn);
int st_difest_isobml(blomgpu_ctx *, int m, int n, int mm, int nn);               // stage_difest_iso.hip: the whole routine
int st_budget_sums(blomgpu_ctx *, int ncall, int n, int nn);
int st_barotp(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_eddtra(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_eddtra_ale(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_cppm(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);     // stage_cppm.hip, called by advect
int st_init_cppm(blomgpu_ctx *);
int st_mxlayr_tail(blomgpu_ctx *, int nn, int k1n);
int ctx_err_words(blomgpu_ctx *);            // allocate err_dev on first use
int ctx_check_errors(blomgpu_ctx *);         // read back all error words, fail with the reference's message
int st_kfpla_halo(blomgpu_ctx *, int n);
int st_cmnfld1(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);   // stage_cmnfld.hip   // phy/mod_cmnfld_routines.F90:1090-1156
int st_cmnfld2(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);   // stage_cmnfld.hip   // phy/mod_cmnfld_routines.F90:1176-1196
int diapfl_column3_launch(blomgpu_ctx *, int n, int nn, int *errflag);
// work-space slots in which k_remap_tile<.., FOLD = true> leaves the new dp, T, S and advected tracers of level kn for pbcor1
// (the slots of the flux planes between k_remap_tile and k_remap_update, remap_common.h: nothing else uses them then)
#define R_BASE(ntr) (8 + 3 * (ntr))
#define R_DP(ntr) (R_BASE(ntr) + 0)
#define R_T(ntr) (R_BASE(ntr) + 1)
#define R_S(ntr) (R_BASE(ntr) + 2)
#define R_TR(ntr, nt) (R_BASE(ntr) + 3 + (nt))
int st_ale_vdifft(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);          // stage_ale_vdiff.hip
int st_ale_vdiffm(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);
int st_ale_forcing(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);         // stage_ale_vdiff.hip
int st_cmnfld_bfsqi_ale(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);    // stage_cmnfld.hip
int st_ale_regrid_remap(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);   // stage_ale.hip
void ale_free(blomgpu_ctx *);
int ale_check_deferred(blomgpu_ctx *);      // stage_ale.hip: the engine's status of the steps since the last check
int st_mxlayr(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);   // stage_mxlayr.hip
int st_mom_pupv(blomgpu_ctx *, int off, int lo, int hi);                          // stage_momtum.hip
int st_convec_velocity(blomgpu_ctx *, int nn);                                    // stage_convec.hip
int launch_dpudpv(blomgpu_ctx *, int off, int flags);                                        // stage_simple.hip
int remap_tile_launch(blomgpu_ctx *, int n, int mm, int nn, int tsel, bool fold);  // stage_remap_tile.hip; tsel 0 all tiles, 1 those that read no halo point, 2 the others
int pbcor_tile_launch(blomgpu_ctx *, int which, int m, int offc, int offf, int from_remap);   // stage_pbcor_tile.hip
int launch_pscan(blomgpu_ctx *, int off, int lo, int hi_off);   // p(k+1)=p(k)+dp(k+off) over lo..ii+hi_off
// xctilr on a device plane stack: `base` points at level lev0 of the field
int st_xctilr(blomgpu_ctx *, double *base, int l1, int ld, int mh, int nh, int itype);
int st_xctilr_arctic_multi(blomgpu_ctx *, int nf, double *const *ptrs, const int *nlevs, const int *mhs, const int *nhs,
                           const int *itypes);
// several stacks (ptrs[f] = first level, nlevs[f] levels) with common widths in one launch where possible
int st_xctilr_multi(blomgpu_ctx *, int nf, double *const *ptrs, const int *nlevs, int mh, int nh, const int *itypes);
int st_crc(blomgpu_ctx *, const double *base, int nlev, int itype, unsigned *crc);
int st_crc_strips(blomgpu_ctx *, double *base, int nlev, int itype, unsigned *out, int cap, int *l0, int *ns);
// locate the field (and level offset) a device pointer belongs to; returns field id or -1
int ctx_locate_ptr(const blomgpu_ctx *, const double *p, size_t *offset);
int rccl_xctilr(blomgpu_ctx *, double *base, int nlev, int mhl, int nhl);   // comm_rccl.hip
int rccl_barotp_replicated(blomgpu_ctx *, int m, int n, int mm, int nn, int k1m, int k1n);   // comm_rccl.hip
blomgpu_ctx *bt_global_ctx(const blomgpu_ctx *);                              // the global barotropic context of a tile, or nullptr
struct RcclLanded {          // received E/W strips left in the transport's buffers, layout [field][level][row][q]
  const double *from_west = nullptr, *from_east = nullptr;
  int has_w = 0, has_e = 0, per = 0, mhl = 0, nhl = 0, nlev = 0;
  double *send_west = nullptr, *send_east = nullptr;   // the send strips (same layout), for a producer that packs them itself
  int prepacked = 0;          // in: the send strips are already packed, skip the pack launch (and phase 1)
};
int rccl_xctilr_multi_ex(blomgpu_ctx *, double *const *fields, int nf, int nlev, int mhl, int nhl, RcclLanded *landed);
int rccl_xctilr_multi(blomgpu_ctx *, double *const *fields, int nf, int nlev, int mhl, int nhl);  // one message per neighbour for up to 4 plane stacks

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own
// 4 MB L2, so with the natural order the rows j-1, j, j+1 of a stencil land in three different L2s and are
// fetched from HBM/Infinity Cache three times.  Re-number the blocks so that XCD x walks a CONTIGUOUS
// eighth of the (layer, row) space: a bijection of [0, gridDim.x*gridDim.y), returned as (bx, by).
__device__ inline void xcd_block(unsigned &bx, unsigned &by) {
  const unsigned gx = gridDim.x, n = gx * gridDim.y;
  const unsigned lin = blockIdx.y * gx + blockIdx.x;
  const unsigned x = lin & 7u, s = lin >> 3;
  const unsigned q = n >> 3, r = n & 7u;
  const unsigned nl = x * q + (x < r ? x : r) + s;
  bx = nl % gx;
  by = nl / gx;
}

// Column kernels (one thread per column, levels walked serially) have one dependent load in flight per thread, and with
// ~1700 wavefronts on the chip that is half the bandwidth a copy reaches (2.8 against 5.9 TB/s,
// profiles/r02_fetch_calibration.txt: k_column8 / k_column8_u<4>).  COLUMN_U levels' loads are therefore issued
// before the first of them is used; the arithmetic and its order are unchanged.
#define COLUMN_U 8
// dst(k+1) = dst(k) + src(k), k = 0..kk-1; src, dst point at level 0 of this thread's column; returns the last sum
__device__ inline double column_scan(double acc, const double *__restrict__ src, double *__restrict__ dst, size_t np, int kk) {
  int k = 0;
  for (; k + COLUMN_U <= kk; k += COLUMN_U) {
    double v[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) v[u] = src[(size_t)(k + u) * np];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) { acc = acc + v[u]; dst[(size_t)(k + u + 1) * np] = acc; }
  }
  for (; k < kk; k++) { acc = acc + src[(size_t)k * np]; dst[(size_t)(k + 1) * np] = acc; }
  return acc;
}

// launch helpers: 1 thread per point of the padded plane, blockIdx.y = level
static inline dim3 plane_grid(const DevView &h, int nlev = 1, int block = 256) {
  return dim3((h.nplane + block - 1) / block, nlev, 1);
}
