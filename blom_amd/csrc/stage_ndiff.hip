// ndiff -- neutral diffusion of tracers for the ALE coordinates (ltedtp = 'neutral', the reference's default with
// vcoord_type = 'cntiso_hybrid'): phy/mod_ndiff.F90, called from inside ale_regrid_remap's j-slice loop
// (phy/mod_ale_regrid_remap.F90:1639-1680) between the regridding and the remapping, on the reconstructed source profiles.
//
// For every pair of neighbouring columns (a u- or a v-face) the reference searches both columns from the surface for
// "neutral interfaces" -- depths in the neighbour where the locally referenced density difference vanishes -- first anchored at
// the source interfaces (search_loop1, a Newton iteration per crossing), then including the destination interfaces
// (search_loop2); between two consecutive neutral interfaces that lie in one source layer and one destination layer on both
// sides a diffusive flux of each tracer is formed from the layer means of the reconstruction polynomials and added to the flux
// convergence of the two destination layers.  The flux convergence is applied to the remapped tracers.
//
// The order in which a cell's flux convergence receives its contributions is fixed by the j-slice loop: the v-face j (from the
// iteration before), the u-faces i and i+1, the v-face j+1; every face adds one term per neutral layer, in search order, and a
// running sum cannot be split.  So the searches run in parallel, ONE THREAD PER FACE, and RECORD their fluxes -- per neutral
// layer the two destination layers and one flux per field (a NaN where the reference's sign tests withhold the flux) --; a
// second kernel, one thread per cell, replays the records of its four faces in the reference's order.  A face also writes its
// own outputs: the fluxes per velocity-point layer (utflld.., utflx..) and the neutral slope (nslpx, nslpy) that
// cmnfld_nnslope_ale and eddtra_ale consume.  (Records per face: every neutral layer consumes a source or a destination
// interface of one of the columns, so 6 kk bounds their number; the space is allocated for that.)
//
// Kernels
//   k_ndiff_prep   ndiff_prep_jslice :959-1026: deepest source/destination layers with mass, the interface values of T and S,
//                  drho/dT, drho/dS at both interfaces of every source layer; zeroes the face fluxes of the ring it covers
//   k_ndiff_flux   ndiff_uflx_jslice / ndiff_vflx_jslice / ndiff_flx :166-953, one thread per u-face (blockIdx.y 0) or v-face (1)
//   k_ndiff_apply  the additions of :876-913 to flxconv_js in the order of the j-slice loop, one thread per cell
// ndiff_update_trc_jslice (:1149-1175) is fused into the copy-back of the remapped fields (stage_ale.hip).
// Roofline: latency of dependent loads along a data-dependent search (two columns, ~6 kk steps); HBM bytes are secondary.
#include "blomgpu_internal.h"
#include "eos.h"
#include "stage_ndiff.h"

#define ND_GRAV 9.806
#define ND_ALPHA0 1.e-3
#define ND_EPSILP 1.e-12
#define ND_ONEMM 9.806
#define ND_MVAL 1.e30
#define ND_DSTSNP_FAC .01   // :41-44
#define ND_RHO_EPS 1.e-5
#define ND_DP_EPS 1.e-5

namespace {

// drhodt, drhods (phy/mod_eos.F90:220-241, :284-304)
__device__ inline double nd_drhodt(double p, double th, double s) {
  using namespace eos;
  const double r1 = a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p;
  const double r2i = 1. / (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p);
  return (a12 + 2. * a14 * th + a15 * s + b12 * p - (a22 + 2. * a24 * th + a25 * s + b22 * p) * r1 * r2i) * r2i;
}
__device__ inline double nd_drhods(double p, double th, double s) {
  using namespace eos;
  const double r1 = a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s + (b11 + b12 * th + b13 * s) * p;
  const double r2i = 1. / (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s + (b21 + b22 * th + b23 * s) * p);
  return (a13 + a15 * th + 2. * a16 * s + b13 * p - (a23 + a25 * th + 2. * a26 * s + b23 * p) * r1 * r2i) * r2i;
}

struct Pc5 { double c1, c2, c3, c4, c5; };

// the five coefficients of layer ks of field nt of column col (the engine extracts npc per layer; the others are zero, as in
// the reference's tpc_src of p_ord + 1 = 5 rows, phy/mod_ale_regrid_remap.F90:1495-1500)
__device__ inline Pc5 nd_pc(const NdArgs &A, size_t np, size_t col, int ks, int nt) {
  const double *b = A.tpc + col + ((size_t)nt * A.npc * A.kk + (size_t)(ks - 1) * A.npc) * np;
  Pc5 r;
  r.c1 = b[0];
  r.c2 = A.npc > 1 ? b[np] : 0.;
  r.c3 = A.npc > 2 ? b[2 * np] : 0.;
  r.c4 = A.npc > 3 ? b[3 * np] : 0.;
  r.c5 = A.npc > 4 ? b[4 * np] : 0.;
  return r;
}
// peval (:62-75), pmeval (:77-103), peval0 / peval1 (phy/mod_ale_regrid_remap.F90:141-159)
__device__ inline double nd_peval(const Pc5 &c, double x) { return (((c.c5 * x + c.c4) * x + c.c3) * x + c.c2) * x + c.c1; }
__device__ inline double nd_pmeval(const Pc5 &c, double x0, double x1) {
  const double c1_2 = 1. / 2., c1_3 = 1. / 3., c1_4 = 1. / 4., c1_5 = 1. / 5.;
  const double b5 = c1_5 * c.c5;
  const double b4 = b5 * x1 + c1_4 * c.c4;
  const double b3 = b4 * x1 + c1_3 * c.c3;
  const double b2 = b3 * x1 + c1_2 * c.c2;
  const double b1 = b2 * x1 + c.c1;
  return (((b5 * x0 + b4) * x0 + b3) * x0 + b2) * x0 + b1;
}
__device__ inline double nd_peval1(const Pc5 &c) { return c.c1 + c.c2 + c.c3 + c.c4 + c.c5; }
// the value of a field at a neutral interface: kind 0 = the polynomial at x, 1 / 2 = the layer's upper / lower interface value
__device__ inline double nd_tni(const Pc5 &c, int kind, double x) { return kind == 0 ? nd_peval(c, x) : (kind == 1 ? c.c1 : nd_peval1(c)); }

// drhoroot, :105-152
__device__ inline double nd_drhoroot(const Pc5 &t, const Pc5 &s, double tf, double sf, double drhodt_l, double drhodt_u, double drhods_l,
                                     double drhods_u) {
  const double eps = 1.e-14, x_tol = 1.e-4;
  double x = .5;
  const double ddrdtdx = drhodt_l - drhodt_u, ddrdsdx = drhods_l - drhods_u;
  for (int n = 1; n <= 10; n++) {
    const double dt = tf - (t.c1 + (t.c2 + (t.c3 + (t.c4 + t.c5 * x) * x) * x) * x);
    const double ds = sf - (s.c1 + (s.c2 + (s.c3 + (s.c4 + s.c5 * x) * x) * x) * x);
    const double drdt = drhodt_l * x + drhodt_u * (1. - x);
    const double drds = drhods_l * x + drhods_u * (1. - x);
    const double dtdx = -(t.c2 + (2. * t.c3 + (3. * t.c4 + 4. * t.c5 * x) * x) * x);
    const double dsdx = -(s.c2 + (2. * s.c3 + (3. * s.c4 + 4. * s.c5 * x) * x) * x);
    const double dr = drdt * dt + drds * ds;
    const double ddrdx = ddrdtdx * dt + drdt * dtdx + ddrdsdx * ds + drds * dsdx;
    const double x_old = x;
    x = fmax2(0., fmin2(1., x_old - dr / copysign(fmax2(eps, fabs(ddrdx)), ddrdx)));
    if (fabs(x - x_old) < x_tol) return x;
  }
  return x;
}

// ndiff_flx, :166-953, for the face between the columns cm ("minus": i-1 or j-1) and cp ("plus", the face's own index).
// sc: the face's scratch column (plane stride nf = 2 nplane); face: its index in the record arrays
// NWS, NWP: 64-bit words of the layers' stability flags (kk bits) and of the set-flags of p_ni_srcdi (2 kk bits)
template <int NWS, int NWP>
__device__ void nd_face(const DevView &V, const NdArgs &A, size_t cm, size_t cp, bool isv, size_t face, double *sc, int *errw) {
  const size_t np = V.nplane, nf = 2 * V.nplane;
  const bool wedge = true;
  int nrec = 0;
  const int kk = V.kk;
  // 1-based accessors in the reference's names
#define PSM(is_, ks_) A.psrc[cm + (size_t)((ks_) + (is_)-2) * np]                       /* p_srcdi_m(is,ks) = p_src(ks+is-1) */
#define PSP(is_, ks_) A.psrc[cp + (size_t)((ks_) + (is_)-2) * np]
#define TSM(is_, ks_, q_) A.tsd[cm + ((size_t)((q_)*kk + (ks_)-1) * 2 + (is_)-1) * np]   /* t_srcdi_m(is,ks,it|is) */
#define TSP(is_, ks_, q_) A.tsd[cp + ((size_t)((q_)*kk + (ks_)-1) * 2 + (is_)-1) * np]
#define DTM(is_, ks_) A.drt[cm + ((size_t)((ks_)-1) * 2 + (is_)-1) * np]
#define DTP(is_, ks_) A.drt[cp + ((size_t)((ks_)-1) * 2 + (is_)-1) * np]
#define DSM(is_, ks_) A.drs[cm + ((size_t)((ks_)-1) * 2 + (is_)-1) * np]
#define DSP(is_, ks_) A.drs[cp + ((size_t)((ks_)-1) * 2 + (is_)-1) * np]
#define PDM(k_) A.pdst[cm + (size_t)((k_)-1) * np]
#define PDP(k_) A.pdst[cp + (size_t)((k_)-1) * np]
#define PNM(is_, ks_) sc[((size_t)((ks_)-1) * 2 + (is_)-1) * nf]                          /* p_ni_srcdi_m */
#define PNP(is_, ks_) sc[((size_t)(2 * kk) + ((ks_)-1) * 2 + (is_)-1) * nf]
#define SNM(k_) sc[((size_t)(4 * kk) + (k_)-1) * nf]                                      /* p_dstsnp_m */
#define SNP(k_) sc[((size_t)(4 * kk) + (kk + 1) + (k_)-1) * nf]
#define NSL(n_) sc[((size_t)(4 * kk) + 2 * (kk + 1) + (n_)-1) * nf]                       /* nslp_src */
#define PNS(n_) sc[((size_t)(4 * kk) + 6 * (kk + 1) + (n_)-1) * nf]                       /* p_nslp_src */
#define DRHO(ism, ksm, isp_, ksp)                                                                                             \
  ((.5 * (DTM(ism, ksm) + DTP(isp_, ksp))) * (TSP(isp_, ksp, 0) - TSM(ism, ksm, 0)) +                                        \
   (.5 * (DSM(ism, ksm) + DSP(isp_, ksp))) * (TSP(isp_, ksp, 1) - TSM(ism, ksm, 1)))
  const int ksmx_m = A.ksmx[cm], ksmx_p = A.ksmx[cp], kdmx_m = A.kdmx[cm], kdmx_p = A.kdmx[cp];
  const double cnslp = ND_ALPHA0 * (isv ? V.f[F_scvyi][cp] : V.f[F_scuxi][cp]) / ND_GRAV;      // :1080, :1135
  unsigned long long stm[NWS], stp[NWS];                                   // stab_src_m, stab_src_p (kk <= 64 NWS)
  for (int q = 0; q < NWS; q++) stm[q] = stp[q] = 0ull;
#define SWI(ks_) (NWS == 1 ? 0 : ((ks_)-1) >> 6)
#define STM(ks_) ((stm[SWI(ks_)] >> (((ks_)-1) & 63)) & 1ull)
#define STP(ks_) ((stp[SWI(ks_)] >> (((ks_)-1) & 63)) & 1ull)
#define STM_SET(ks_) stm[SWI(ks_)] |= 1ull << (((ks_)-1) & 63)
#define STP_SET(ks_) stp[SWI(ks_)] |= 1ull << (((ks_)-1) & 63)
  // which entries of p_ni_srcdi_m, p_ni_srcdi_p have been set (the reference tests them against mval): the index walks of the
  // second search then need no loads
  unsigned long long pbm[NWP], pbp[NWP];
  for (int q = 0; q < NWP; q++) pbm[q] = pbp[q] = 0ull;
#define PBI(is_, ks_) (((ks_)-1) * 2 + (is_)-1)
#define PWI(is_, ks_) (NWP == 1 ? 0 : PBI(is_, ks_) >> 6)
#define PBM(is_, ks_) ((pbm[PWI(is_, ks_)] >> (PBI(is_, ks_) & 63)) & 1ull)
#define PBP(is_, ks_) ((pbp[PWI(is_, ks_)] >> (PBI(is_, ks_) & 63)) & 1ull)
#define PNM_SET(is_, ks_, v_) do { PNM(is_, ks_) = (v_); pbm[PWI(is_, ks_)] |= 1ull << (PBI(is_, ks_) & 63); } while (0)
#define PNP_SET(is_, ks_, v_) do { PNP(is_, ks_) = (v_); pbp[PWI(is_, ks_)] |= 1ull << (PBI(is_, ks_) & 63); } while (0)
  for (int k = 1; k <= kk; k++) { PNM(1, k) = ND_MVAL; PNM(2, k) = ND_MVAL; PNP(1, k) = ND_MVAL; PNP(2, k) = ND_MVAL; }
  int nns = 0;
  int is_m, is_p, ks_m, ks_p, kssa_m = 0, kssa_p = 0;
  double p_ni_m_prev, p_ni_p_prev, drho_curr = 0., pml = 0.;
  // ---- first search: neutral interfaces anchored at the source interfaces, :225-392 -------------------------------------------
  if (A.surface_align) {
    gcd_t dpml = V.f[F_dpml];
    pml = .5 * (PSM(1, 1) + dpml[cm] + PSP(1, 1) + dpml[cp]);
    kssa_m = 2;
    while (kssa_m <= ksmx_m) {
      if (PSM(1, kssa_m) > pml) break;
      kssa_m = kssa_m + 1;
    }
    kssa_p = 2;
    while (kssa_p <= ksmx_p) {
      if (PSP(1, kssa_p) > pml) break;
      kssa_p = kssa_p + 1;
    }
    is_m = 1; ks_m = kssa_m; is_p = 1; ks_p = kssa_p;
    p_ni_m_prev = pml; p_ni_p_prev = pml;
  } else {
    is_m = 1; ks_m = 1; is_p = 1; ks_p = 1;
    p_ni_m_prev = PSM(1, 1); p_ni_p_prev = PSP(1, 1);
  }
  // The reference's loop classifies the density difference at the current pair of interfaces and then walks down column m
  // and / or column p until the difference has moved on by rho_eps.  Here that is ONE loop in which every pass moves one
  // interface in one of the columns (mode 1: column m, 2: column p; 0: classify first): a pass has a single round of loads
  // whichever column the lanes of a wavefront are walking.  The values at the current interfaces, and those at the upper
  // interface of the current layers, stay in registers.
  double psm = 0., tsm = 0., ssm = 0., dtm = 0., dsm = 0., psm1 = 0., dtm1 = 0., dsm1 = 0.;
  double psp = 0., tsp = 0., ssp = 0., dtp = 0., dsp = 0., psp1 = 0., dtp1 = 0., dsp1 = 0.;
  double lastm2 = 0., lastp2 = 0., prevm2 = 0., prevp2 = 0.;      // p_ni_srcdi(2, ks) as last written, and that of the layer above
#define DRHO_CUR() ((.5 * (dtm + dtp)) * (tsp - tsm) + (.5 * (dsm + dsp)) * (ssp - ssm))
  if (ks_m <= ksmx_m && ks_p <= ksmx_p) {
    psm = PSM(is_m, ks_m); tsm = TSM(is_m, ks_m, 0); ssm = TSM(is_m, ks_m, 1); dtm = DTM(is_m, ks_m); dsm = DSM(is_m, ks_m);
    psp = PSP(is_p, ks_p); tsp = TSP(is_p, ks_p, 0); ssp = TSP(is_p, ks_p, 1); dtp = DTP(is_p, ks_p); dsp = DSP(is_p, ks_p);
    psm1 = psm; dtm1 = dtm; dsm1 = dsm; psp1 = psp; dtp1 = dtp; dsp1 = dsp;                // (both start at an upper interface)
    drho_curr = DRHO_CUR();
  }
  int mode = 0;
  bool then_p = false;
  while (ks_m <= ksmx_m && ks_p <= ksmx_p) {                    // search_loop1
    if (mode == 0) {
      const bool drho_neg = drho_curr <= -ND_RHO_EPS, drho_pos = drho_curr >= ND_RHO_EPS;
      const bool drho_zero = !(drho_neg || drho_pos);
      if (is_m + ks_m > 2 && is_p + ks_p > 2) {
        if (drho_neg) {
          if (is_m == 2) {
            const double drhodt_x0 = .5 * (dtm1 + dtp), drhodt_x1 = .5 * (dtm + dtp);
            const double drhods_x0 = .5 * (dsm1 + dsp), drhods_x1 = .5 * (dsm + dsp);
            const double x_ni = nd_drhoroot(nd_pc(A, np, cm, ks_m, 0), nd_pc(A, np, cm, ks_m, 1), tsp, ssp, drhodt_x1, drhodt_x0, drhods_x1, drhods_x0);
            const double p_ni = psm * x_ni + psm1 * (1. - x_ni);
            if (p_ni > p_ni_m_prev) {
              p_ni_m_prev = p_ni;
              PNP_SET(is_p, ks_p, p_ni);
              if (is_p == 2) lastp2 = p_ni;
              nns = nns + 1;
              if (wedge) { NSL(nns) = -cnslp * (psp - p_ni); PNS(nns) = .5 * (psp + p_ni); }
            }
          }
        } else if (drho_pos) {
          if (is_p == 2) {
            const double drhodt_x0 = .5 * (dtm + dtp1), drhodt_x1 = .5 * (dtm + dtp);
            const double drhods_x0 = .5 * (dsm + dsp1), drhods_x1 = .5 * (dsm + dsp);
            const double x_ni = nd_drhoroot(nd_pc(A, np, cp, ks_p, 0), nd_pc(A, np, cp, ks_p, 1), tsm, ssm, drhodt_x1, drhodt_x0, drhods_x1, drhods_x0);
            const double p_ni = psp * x_ni + psp1 * (1. - x_ni);
            if (p_ni > p_ni_p_prev) {
              p_ni_p_prev = p_ni;
              PNM_SET(is_m, ks_m, p_ni);
              if (is_m == 2) lastm2 = p_ni;
              nns = nns + 1;
              if (wedge) { NSL(nns) = -cnslp * (p_ni - psm); PNS(nns) = .5 * (p_ni + psm); }
            }
          }
        } else {
          PNP_SET(is_p, ks_p, psm);
          PNM_SET(is_m, ks_m, psp);
          if (is_p == 2) lastp2 = psm;
          if (is_m == 2) lastm2 = psp;
          nns = nns + 1;
          if (wedge) { NSL(nns) = -cnslp * (psp - psm); PNS(nns) = .5 * (psp + psm); }
        }
      }
      if (drho_zero || drho_pos) { mode = 1; then_p = drho_zero || drho_neg; }
      else mode = 2;
    }
    const double drho_prev = drho_curr;
    const bool wm = mode == 1;
    if (wm) {
      if (is_m == 1) is_m = 2;
      else {
        ks_m = ks_m + 1;
        if (ks_m > ksmx_m) break;
        is_m = 1;
        prevm2 = lastm2;
      }
    } else {
      if (is_p == 1) is_p = 2;
      else {
        ks_p = ks_p + 1;
        if (ks_p > ksmx_p) break;
        is_p = 1;
        prevp2 = lastp2;
      }
    }
    {
      const size_t col = wm ? cm : cp;
      const int is_ = wm ? is_m : is_p, ks_ = wm ? ks_m : ks_p;
      const double v_ps = A.psrc[col + (size_t)(ks_ + is_ - 2) * np], v_t = A.tsd[col + ((size_t)(ks_ - 1) * 2 + is_ - 1) * np],
                   v_s = A.tsd[col + ((size_t)(kk + ks_ - 1) * 2 + is_ - 1) * np], v_dt = A.drt[col + ((size_t)(ks_ - 1) * 2 + is_ - 1) * np],
                   v_ds = A.drs[col + ((size_t)(ks_ - 1) * 2 + is_ - 1) * np];
      if (wm) {
        psm = v_ps; tsm = v_t; ssm = v_s; dtm = v_dt; dsm = v_ds;
        if (is_ == 1) { psm1 = v_ps; dtm1 = v_dt; dsm1 = v_ds; }
      } else {
        psp = v_ps; tsp = v_t; ssp = v_s; dtp = v_dt; dsp = v_ds;
        if (is_ == 1) { psp1 = v_ps; dtp1 = v_dt; dsp1 = v_ds; }
      }
    }
    drho_curr = DRHO_CUR();
    if (wm) {
      if (drho_prev - drho_curr > ND_RHO_EPS) {
        if (is_m == 2 && psm - psm1 > ND_ONEMM) STM_SET(ks_m);
        mode = then_p ? 2 : 0;
      } else if (is_m == 1 && PBM(2, ks_m - 1)) {                // (a copy of mval leaves the entry unset)
        PNM_SET(1, ks_m, prevm2);
      }
    } else {
      if (drho_curr - drho_prev > ND_RHO_EPS) {
        if (is_p == 2 && psp - psp1 > ND_ONEMM) STP_SET(ks_p);
        mode = 0;
      } else if (is_p == 1 && PBP(2, ks_p - 1)) {
        PNP_SET(1, ks_p, prevp2);
      }
    }
  }
#undef DRHO_CUR
#ifndef BLOM_HOSTEMU
  if (A.prof && threadIdx.x == 0 && 6 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 5 < (size_t)A.prof_words) A.prof[6 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 3] = wall_clock64();
#endif
  // ---- alignment with the surface above the uppermost neutral interface, :394-464 ---------------------------------------------
  if (A.surface_align) {
    int issa_m = 1, issa_p = 1;
    while (kssa_m <= ksmx_m) {
      if (PBM(issa_m, kssa_m)) break;
      if (issa_m == 1) issa_m = 2;
      else { kssa_m = kssa_m + 1; issa_m = 1; }
    }
    while (kssa_p <= ksmx_p) {
      if (PBP(issa_p, kssa_p)) break;
      if (issa_p == 1) issa_p = 2;
      else { kssa_p = kssa_p + 1; issa_p = 1; }
    }
    if (kssa_m > ksmx_m || kssa_p > ksmx_p) {
      PNM_SET(1, 1, PSM(1, 1));
      for (ks_m = 1; ks_m <= ksmx_m - 1; ks_m++) {
        if (PSM(1, ks_m) > PSP(2, ksmx_p)) break;
        const double p_ni = fmin2(PSM(2, ks_m), PSP(2, ksmx_p));
        PNM_SET(1, ks_m + 1, p_ni);
        PNM_SET(2, ks_m, p_ni);
        STM_SET(ks_m);
      }
      PNP_SET(1, 1, PSP(1, 1));
      for (ks_p = 1; ks_p <= ksmx_p - 1; ks_p++) {
        if (PSP(1, ks_p) > PSM(2, ksmx_m)) break;
        const double p_ni = fmin2(PSP(2, ks_p), PSM(2, ksmx_m));
        PNP_SET(1, ks_p + 1, p_ni);
        PNP_SET(2, ks_p, p_ni);
        STP_SET(ks_p);
      }
    } else {
      double p1_m, p2_m, p1_p, p2_p;
      if (PSM(issa_m, kssa_m) < PNP(issa_p, kssa_p)) {
        p1_m = PSM(1, 1); p2_m = PSM(issa_m, kssa_m); p1_p = PSP(1, 1); p2_p = PNM(issa_m, kssa_m);
      } else {
        p1_m = PSM(1, 1); p2_m = PNP(issa_p, kssa_p); p1_p = PSP(1, 1); p2_p = PSP(issa_p, kssa_p);
      }
      PNM_SET(1, 1, p1_p);
      for (ks_m = 1; ks_m <= kssa_m - 1; ks_m++) {
        const double p_ni = ((PSM(2, ks_m) - p1_m) * p2_p + (p2_m - PSM(2, ks_m)) * p1_p) / (p2_m - p1_m);
        PNM_SET(1, ks_m + 1, p_ni);
        PNM_SET(2, ks_m, p_ni);
        STM_SET(ks_m);
      }
      PNP_SET(1, 1, p1_m);
      for (ks_p = 1; ks_p <= kssa_p - 1; ks_p++) {
        const double p_ni = ((PSP(2, ks_p) - p1_p) * p2_m + (p2_p - PSP(2, ks_p)) * p1_m) / (p2_p - p1_p);
        PNP_SET(1, ks_p + 1, p_ni);
        PNP_SET(2, ks_p, p_ni);
        STP_SET(ks_p);
      }
    }
  }
  // ---- destination interfaces snapped to the source interfaces they nearly coincide with, :476-508 ----------------------------
  {
    SNM(1) = PDM(1);
    double dp_dst_u = PDM(2) - PDM(1);
    const int km = ksmx_m < kdmx_m ? ksmx_m : kdmx_m;
    for (int k = 2; k <= km; k++) {
      const double dp_dst_l = PDM(k + 1) - PDM(k);
      SNM(k) = fabs(PDM(k) - PSM(1, k)) < fmin2(dp_dst_u, dp_dst_l) * ND_DSTSNP_FAC ? PSM(1, k) : PDM(k);
      dp_dst_u = dp_dst_l;
    }
    for (int k = km + 1; k <= kdmx_m + 1; k++) SNM(k) = PDM(k);
    SNP(1) = PDP(1);
    dp_dst_u = PDP(2) - PDP(1);
    const int kp = ksmx_p < kdmx_p ? ksmx_p : kdmx_p;
    for (int k = 2; k <= kp; k++) {
      const double dp_dst_l = PDP(k + 1) - PDP(k);
      SNP(k) = fabs(PDP(k) - PSP(1, k)) < fmin2(dp_dst_u, dp_dst_l) * ND_DSTSNP_FAC ? PSP(1, k) : PDP(k);
      dp_dst_u = dp_dst_l;
    }
    for (int k = kp + 1; k <= kdmx_p + 1; k++) SNP(k) = PDP(k);
  }
#ifndef BLOM_HOSTEMU
  if (A.prof && threadIdx.x == 0 && 6 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 5 < (size_t)A.prof_words) A.prof[6 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 4] = wall_clock64();
#endif
  // ---- second search, :510-921 ------------------------------------------------------------------------------------------------
  {
    is_m = 2; ks_m = 0; is_p = 2; ks_p = 0;
    int kd_m = 0, kd_p = 0, isn_m = 1, isn_p = 1, ksn_m = 1, ksn_p = 1, ks_m_prev = 0, ks_p_prev = 0;
    bool advance_src_m = true, advance_src_p = true, advance_dst_m = true, advance_dst_p = true;
    // the previous (P) and the current (C) neutral interface: pressures and positions in the two columns, and how the values
    // there are to be taken (nd_tni).  The reference flips two slots; here the current one becomes the previous one.
    double pmP, pmC = 0., ppP, ppC = 0., xmP = 0., xmC = 0., xpP = 0., xpC = 0.;
    int kmP = 0, kmC = 0, kpP = 0, kpC = 0;
    // in registers between index changes: p_srcdi(is,ks), (1,ks), (2,ks), p_ni_srcdi(is,ks), (isn,ksn), p_srcdi(isn,ksn), p_dstsnp(kd+1)
    double sn_m = 0., sn_p = 0., dd_m = 0., dd_p = 0.;      // p_dstsnp(kd), p_dst(kd+1) - p_dst(kd) of the current destination layers
    double a_m = 0., b_m = 0., c_m = 0., d_m = 0., e_m = 0., f_m = 0., g_m = 0., a_p = 0., b_p = 0., c_p = 0., d_p = 0., e_p = 0., f_p = 0., g_p = 0.;
    pmP = -ND_MVAL; ppP = -ND_MVAL;
    while (true) {                                              // search_loop2
      // the index walks first (no loads: the set-bits of p_ni_srcdi and the stability flags are in registers), then every value
      // the new indices need in ONE round of loads
      bool out = false;
      if (advance_src_m) {
        while (true) {
          if (is_m == 1) {
            is_m = 2;
            if (STM(ks_m)) break;
          } else {
            ks_m = ks_m + 1;
            if (ks_m > ksmx_m) { out = true; break; }
            is_m = 1;
            if (STM(ks_m) && PBM(is_m, ks_m)) break;
          }
        }
        if (!out) {
          isn_m = is_m; ksn_m = ks_m;
          while (!PBM(isn_m, ksn_m)) {
            if (isn_m == 1) isn_m = 2;
            else {
              if (ksn_m == ksmx_m) break;
              ksn_m = ksn_m + 1;
              isn_m = 1;
            }
          }
        }
      }
      if (out) break;
      if (advance_src_p) {
        while (true) {
          if (is_p == 1) {
            is_p = 2;
            if (STP(ks_p)) break;
          } else {
            ks_p = ks_p + 1;
            if (ks_p > ksmx_p) { out = true; break; }
            is_p = 1;
            if (STP(ks_p) && PBP(is_p, ks_p)) break;
          }
        }
        if (!out) {
          isn_p = is_p; ksn_p = ks_p;
          while (!PBP(isn_p, ksn_p)) {
            if (isn_p == 1) isn_p = 2;
            else {
              if (ksn_p == ksmx_p) break;
              ksn_p = ksn_p + 1;
              isn_p = 1;
            }
          }
        }
      }
      if (out) break;
      if (advance_dst_m) {
        kd_m = kd_m + 1;
        if (kd_m > kdmx_m) break;
      }
      if (advance_dst_p) {
        kd_p = kd_p + 1;
        if (kd_p > kdmx_p) break;
      }
      if (advance_src_m) {
        d_m = PNM(is_m, ks_m); e_m = PNM(isn_m, ksn_m);
        a_m = PSM(is_m, ks_m); b_m = PSM(1, ks_m); c_m = PSM(2, ks_m); f_m = PSM(isn_m, ksn_m);
      }
      if (advance_src_p) {
        d_p = PNP(is_p, ks_p); e_p = PNP(isn_p, ksn_p);
        a_p = PSP(is_p, ks_p); b_p = PSP(1, ks_p); c_p = PSP(2, ks_p); f_p = PSP(isn_p, ksn_p);
      }
      // (with a destination index its snapped upper interface and the layer's thickness come along in the same round of loads: a pass
      // that finds a neutral interface needs them, and asking then would be a second memory round trip in every such pass)
      if (advance_dst_m) { g_m = SNM(kd_m + 1); sn_m = SNM(kd_m); dd_m = PDM(kd_m + 1) - PDM(kd_m); }
      if (advance_dst_p) { g_p = SNP(kd_p + 1); sn_p = SNP(kd_p); dd_p = PDP(kd_p + 1) - PDP(kd_p); }
      if (pmP == -ND_MVAL) {
        if ((e_m - f_p) < (e_p - f_m)) {
          pmP = f_m;
          ppP = e_m;
        } else {
          pmP = e_p;
          ppP = f_p;
        }
      }
      while (true) {                                            // both columns' destination indices step together
        const bool nm = g_m <= fmax2(b_m, pmP), npp = g_p <= fmax2(b_p, ppP);
        if (!nm && !npp) break;
        if (nm) {
          kd_m = kd_m + 1;
          if (kd_m > kdmx_m) { out = true; break; }
        }
        if (npp) {
          kd_p = kd_p + 1;
          if (kd_p > kdmx_p) { out = true; break; }
        }
        if (nm) { g_m = SNM(kd_m + 1); sn_m = SNM(kd_m); dd_m = PDM(kd_m + 1) - PDM(kd_m); }
        if (npp) { g_p = SNP(kd_p + 1); sn_p = SNP(kd_p); dd_p = PDP(kd_p + 1) - PDP(kd_p); }
      }
      if (out) break;
      advance_src_m = false; advance_src_p = false; advance_dst_m = false; advance_dst_p = false;
      int case_m = 3, case_p = 3;
      if (a_m <= e_p) {
        if (a_m <= g_m) case_m = 1;
      } else if (e_p <= g_m) case_m = 2;
      if (a_p <= e_m) {
        if (a_p <= g_p) case_p = 1;
      } else if (e_m <= g_p) case_p = 2;
      bool found_ni = false;
      if (case_m == 3 && case_p == 3) {
        if (is_p == 2 && is_m == 2) {
          pmC = g_m;
          ppC = g_p;
          const double pu_m = pmP, pu_p = ppP;
          double pl_m, pl_p;
          if ((e_m - f_p) < (e_p - f_m)) {
            pl_m = f_m; pl_p = e_m;
          } else {
            pl_m = e_p; pl_p = f_p;
          }
          const double pp1 = (pmC - pu_m) * (pl_p - pu_p), pp2 = (ppC - pu_p) * (pl_m - pu_m);
          if (fabs(pp1 - pp2) < ND_DP_EPS * fmax2(ND_DP_EPS, pl_m - pu_m + pl_p - pu_p)) {
            advance_dst_m = true; advance_dst_p = true;
          } else if (pp1 < pp2) {
            ppC = pu_p + pp1 / (pl_m - pu_m);
            advance_dst_m = true;
          } else {
            pmC = pu_m + pp2 / (pl_p - pu_p);
            advance_dst_p = true;
          }
          if (pmC >= b_m && pmC <= c_m && ppC >= b_p && ppC <= c_p) {
            xmC = (pmC - b_m) / (c_m - b_m);
            xpC = (ppC - b_p) / (c_p - b_p);
            kmC = 0; kpC = 0;
            found_ni = true;
          }
        } else {
          if (is_p != 2) advance_dst_m = true;
          if (is_m != 2) advance_dst_p = true;
        }
      } else if (case_m == 3) {
        if (is_p == 2) {
          pmC = g_m;
          if (case_p == 1)
            ppC = ppP + (pmC - pmP) * (f_p - ppP) / (e_p - pmP);
          else
            ppC = ppP + (pmC - pmP) * (e_m - ppP) / (f_m - pmP);
          if (ppC >= b_p && ppC <= c_p) {
            xmC = (g_m - b_m) / (c_m - b_m);
            xpC = (ppC - b_p) / (c_p - b_p);
            kmC = 0; kpC = 0;
            found_ni = true;
            advance_dst_m = true;
          } else {
            if (case_p == 1 && d_p == ND_MVAL) advance_src_p = true;
            else advance_dst_m = true;
          }
        } else advance_dst_m = true;
      } else if (case_p == 3) {
        if (is_m == 2) {
          ppC = g_p;
          if (case_m == 1)
            pmC = pmP + (ppC - ppP) * (f_m - pmP) / (e_m - ppP);
          else
            pmC = pmP + (ppC - ppP) * (e_p - pmP) / (f_p - ppP);
          if (pmC >= b_m && pmC <= c_m) {
            xpC = (g_p - b_p) / (c_p - b_p);
            xmC = (pmC - b_m) / (c_m - b_m);
            kmC = 0; kpC = 0;
            found_ni = true;
            advance_dst_p = true;
          } else {
            if (case_m == 1 && d_m == ND_MVAL) advance_src_m = true;
            else advance_dst_p = true;
          }
        } else advance_dst_p = true;
      } else if (case_m == 1 && case_p == 1) {
        if (d_m != ND_MVAL && d_p != ND_MVAL) {
          xmC = (double)(is_m - 1);
          pmC = a_m;
          xpC = (double)(is_p - 1);
          ppC = a_p;
          kmC = is_m; kpC = is_p;
          found_ni = true;
          advance_src_m = true; advance_src_p = true;
        } else {
          if (d_m == ND_MVAL) advance_src_m = true;
          if (d_p == ND_MVAL) advance_src_p = true;
        }
      } else if (case_m == 1) {
        if (d_m != ND_MVAL && d_m >= b_p) {
          xmC = (double)(is_m - 1);
          pmC = a_m;
          ppC = d_m;
          xpC = (ppC - b_p) / (c_p - b_p);
          kmC = is_m; kpC = 0;
          found_ni = true;
        }
        advance_src_m = true;
      } else if (case_p == 1) {
        if (d_p != ND_MVAL && d_p >= b_m) {
          xpC = (double)(is_p - 1);
          ppC = a_p;
          pmC = d_p;
          xmC = (pmC - b_m) / (c_m - b_m);
          kpC = is_p; kmC = 0;
          found_ni = true;
        }
        advance_src_p = true;
      } else {
        advance_src_m = true; advance_src_p = true;
      }
      if (found_ni) {
        const double dp_ni_m = fmin2(pmC - pmP, dd_m);
        const double dp_ni_p = fmin2(ppC - ppP, dd_p);
        const double dp_ni = 2. * dp_ni_m * dp_ni_p / fmax2(dp_ni_m + dp_ni_p, 2. * ND_DP_EPS);
        if (ks_m == ks_m_prev && ks_p == ks_p_prev && pmP >= sn_m && pmC <= g_m && ppP >= sn_p &&
            ppC <= g_p && dp_ni > 2. * ND_DP_EPS) {
          // the record of this neutral layer (k_ndiff_eval forms the fluxes from it): destination and source layers, how the
          // interface values are to be taken, the positions of the two neutral interfaces in the source layers, the thickness
          const bool keep = nrec < A.nrec_max;
          if (!keep) atomicOr(errw, 1);
          if (keep) {
            A.rec_k[face + (size_t)nrec * nf] = kd_m | (kd_p << 16);
            A.rec_s[face + (size_t)nrec * nf] = ks_m | (ks_p << 8) | (kmP << 16) | (kmC << 18) | (kpP << 20) | (kpC << 22);
            double *rg = A.rec_g + face + (size_t)nrec * 7 * nf;
            rg[0] = xmP; rg[nf] = xmC; rg[2 * nf] = xpP; rg[3 * nf] = xpC; rg[4 * nf] = dp_ni;
            rg[5 * nf] = .5 * (pmP + ppP); rg[6 * nf] = .5 * (pmC + ppC);
            nrec = nrec + 1;
          }
        }
        ks_m_prev = ks_m; ks_p_prev = ks_p;
        pmP = pmC; ppP = ppC; xmP = xmC; xpP = xpC; kmP = kmC; kpP = kpC;
      }
    }
  }
  A.rec_n[face] = nrec;
  // ---- neutral slope at the destination interfaces, :923-951 ------------------------------------------------------------------
  if (wedge) {
    gd_t nsl = (isv ? V.f[F_nslpy] : V.f[F_nslpx]);
#define NSXY(kd_) nsl[cp + (size_t)((kd_)-1) * np]
    if (nns == 0) {
      for (int kd = 1; kd <= kk; kd++) NSXY(kd) = 0.;
    } else {
      int kd;
      double p_nslp_dst = 0.;
      for (kd = 1; kd <= kk; kd++) {
        p_nslp_dst = .5 * (PDM(kd) + PDP(kd));
        if (p_nslp_dst > PNS(1)) break;
        NSXY(kd) = NSL(1);
      }
      if (kd <= kk) {
        int ks = 1;
        while (true) {                                          // interp_loop
          bool out = false;
          while (p_nslp_dst > PNS(ks)) {
            if (ks == nns) { out = true; break; }
            ks = ks + 1;
          }
          if (out) break;
          const double q = (PNS(ks) - p_nslp_dst) / fmax2(PNS(ks) - PNS(ks - 1), ND_EPSILP);
          NSXY(kd) = q * NSL(ks - 1) + (1. - q) * NSL(ks);
          kd = kd + 1;
          if (kd > kk) break;
          p_nslp_dst = .5 * (PDM(kd) + PDP(kd));
        }
        for (; kd <= kk; kd++) NSXY(kd) = NSL(nns);
      }
    }
#undef NSXY
  }
}

}  // namespace

// ndiff_prep_jslice, :959-1026, for the p-columns of the ring (i = 0..ii+1, j = 0..jj+1)
__global__ void k_ndiff_prep(const DevView *__restrict__ Vp, NdArgs A, int *__restrict__ ksmx, int *__restrict__ kdmx, double *__restrict__ tsd,
                             double *__restrict__ drt, double *__restrict__ drs) {
  const DevView &V = *Vp;
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  const size_t c = t_, np = V.nplane;
  const int kk = V.kk, mm = A.mm;
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1) return;
  if (V.m[I_ip][c]) {
    // deepest source layer with mass (reconstruct_trc_jslice, phy/mod_ale_regrid_remap.F90:218-222) and destination layer (:981-986)
    int ks = kk, kd = kk;
    const double pbs = A.psrc[c + (size_t)kk * np], pbd = A.pdst[c + (size_t)kk * np];
    for (int k = kk; k >= 1; k--) {
      if (A.psrc[c + (size_t)(k - 1) * np] == pbs) ks = k - 1;
      if (A.pdst[c + (size_t)(k - 1) * np] == pbd) kd = k - 1;
    }
    ksmx[c] = ks;
    kdmx[c] = kd;
    for (int k = 1; k <= ks; k++) {
      const Pc5 t = nd_pc(A, np, c, k, 0), s = nd_pc(A, np, c, k, 1);
      const double t1 = t.c1, t2 = nd_peval1(t), s1 = s.c1, s2 = nd_peval1(s);      // t_srcdi, phy/mod_ale_regrid_remap.F90:252-255
      const double p1 = A.psrc[c + (size_t)(k - 1) * np], p2 = A.psrc[c + (size_t)k * np];
      tsd[c + ((size_t)(k - 1) * 2) * np] = t1;
      tsd[c + ((size_t)(k - 1) * 2 + 1) * np] = t2;
      tsd[c + ((size_t)(kk + k - 1) * 2) * np] = s1;
      tsd[c + ((size_t)(kk + k - 1) * 2 + 1) * np] = s2;
      drt[c + ((size_t)(k - 1) * 2) * np] = nd_drhodt(p1, t1, s1);
      drt[c + ((size_t)(k - 1) * 2 + 1) * np] = nd_drhodt(p2, t2, s2);
      drs[c + ((size_t)(k - 1) * 2) * np] = nd_drhods(p1, t1, s1);
      drs[c + ((size_t)(k - 1) * 2 + 1) * np] = nd_drhods(p2, t2, s2);
    }
    for (int n = 0; n < kk * A.ntr_loc; n++) A.flx[c + (size_t)n * np] = 0.;
  }
  for (int k = 0; k <= kk; k++) {
    A.puv[c + (size_t)k * np] = V.f[F_pu][c + (size_t)k * np];
    A.puv[c + (size_t)(kk + 1 + k) * np] = V.f[F_pv][c + (size_t)k * np];
  }
  if (V.m[I_iu][c])
    for (int k = 0; k < kk; k++) { V.f[F_utflld][c + (size_t)(k + mm) * np] = 0.; V.f[F_usflld][c + (size_t)(k + mm) * np] = 0.; }
  if (V.m[I_iv][c])
    for (int k = 0; k < kk; k++) { V.f[F_vtflld][c + (size_t)(k + mm) * np] = 0.; V.f[F_vsflld][c + (size_t)(k + mm) * np] = 0.; }
}

// one thread per u-face (rows 1..jj, i = 1..ii+1: ndiff_uflx_jslice) or v-face (rows 1..jj+1, i = 1..ii: ndiff_vflx_jslice)
template <int NWS, int NWP>
__global__ __launch_bounds__(64) void k_ndiff_flux(const DevView *__restrict__ Vp, NdArgs A, int *__restrict__ errw) {
  const DevView &V = *Vp;
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  const size_t c = t_;
  const bool isv = blockIdx.y == 1;
  const size_t face = c + (isv ? V.nplane : 0);
  const bool on = isv ? (j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii && V.m[I_iv][c]) : (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1 && V.m[I_iu][c]);
#ifndef BLOM_HOSTEMU
  long long t0 = 0;
  if (A.prof) t0 = wall_clock64();
#endif
  if (!on) A.rec_n[face] = 0;
  else nd_face<NWS, NWP>(V, A, isv ? c - V.ni : c - 1, c, isv, face, A.scr + face, errw);
#ifndef BLOM_HOSTEMU
  if (A.prof && 6 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 5 < (size_t)A.prof_words) {
    const long long t1 = wall_clock64();
    const size_t w = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) { A.prof[6 * w] = t0; A.prof[6 * w + 1] = t1; A.prof[6 * w + 2] = 0; }
    atomicMax((unsigned long long *)&A.prof[6 * w + 2], (unsigned long long)A.rec_n[face]);
  }
#endif
}

// the fluxes of a face's records, :860-913.  The records do not depend on each other: blockIdx.y strides over them; heat and
// salt share one sign test, every tracer has its own.  A withheld flux is a NaN in the record.
#define ND_EVAL_RY 16
#define ND_RB 8            // records loaded ahead by the kernels that walk a face's records in order
__global__ __launch_bounds__(64) void k_ndiff_eval(const DevView *__restrict__ Vp, NdArgs A) {
  const DevView &V = *Vp;
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t np = V.nplane, nf = 2 * np;
  if ((size_t)t_ >= nf) return;
  const size_t face = t_;
  const int n = A.rec_n[face];
  if ((int)blockIdx.y >= n) return;
  const bool isv = face >= np;
  const size_t cp = isv ? face - np : face, cm = isv ? cp - V.ni : cp - 1;
  const int kk = V.kk, nn = A.nn, ntr_loc = A.ntr_loc;
  const double cdiff = isv ? V.P.delt1 * V.f[F_scvx][cp] * V.f[F_scvyi][cp] : V.P.delt1 * V.f[F_scuy][cp] * V.f[F_scuxi][cp];   // :1079, :1134
  gcd_t difiso = V.f[F_difiso];
  const double withheld = __builtin_nan("");
  // A flux that passes the sign tests but is itself a NaN (a NaN in difiso or in the polynomial coefficients: the state has blown up)
  // must not look like a withheld one, or it would be dropped where the reference carries it into the tracers: it travels as an
  // infinity, which the sums turn into the non-finite state the reference would show
  auto flux = [&](bool pass, double f) { return pass ? (f == f ? f : __builtin_inf()) : withheld; };
  for (int r = blockIdx.y; r < n; r += ND_EVAL_RY) {
    const int rs = A.rec_s[face + (size_t)r * nf];
    const int ks_m = rs & 255, ks_p = (rs >> 8) & 255, km0 = (rs >> 16) & 3, km1 = (rs >> 18) & 3, kp0 = (rs >> 20) & 3, kp1 = (rs >> 22) & 3;
    const double *rg = A.rec_g + face + (size_t)r * 7 * nf;
    const double xm0 = rg[0], xm1 = rg[nf], xp0 = rg[2 * nf], xp1 = rg[3 * nf], dp_ni = rg[4 * nf];
    double *rf = A.rec_f + face + (size_t)r * ntr_loc * nf;
    const double q = .5 * cdiff * (difiso[cm + (size_t)(ks_m - 1) * np] + difiso[cp + (size_t)(ks_p - 1) * np]) * dp_ni;
    const size_t om = cm + (size_t)(ks_m - 1 + nn) * np, op = cp + (size_t)(ks_p - 1 + nn) * np;
    {
      const Pc5 tm = nd_pc(A, np, cm, ks_m, 0), tp = nd_pc(A, np, cp, ks_p, 0), sm = nd_pc(A, np, cm, ks_m, 1), sp = nd_pc(A, np, cp, ks_p, 1);
      const double dt = nd_pmeval(tm, xm0, xm1) - nd_pmeval(tp, xp0, xp1);
      const double ds = nd_pmeval(sm, xm0, xm1) - nd_pmeval(sp, xp0, xp1);
      const bool pass = dt * (V.f[F_temp][om] - V.f[F_temp][op]) >= 0. && dt * (nd_tni(tm, km0, xm0) - nd_tni(tp, kp0, xp0)) >= 0. &&
                        dt * (nd_tni(tm, km1, xm1) - nd_tni(tp, kp1, xp1)) >= 0. && ds * (V.f[F_saln][om] - V.f[F_saln][op]) >= 0. &&
                        ds * (nd_tni(sm, km0, xm0) - nd_tni(sp, kp0, xp0)) >= 0. && ds * (nd_tni(sm, km1, xm1) - nd_tni(sp, kp1, xp1)) >= 0.;
      rf[0] = flux(pass, q * dt);
      rf[nf] = flux(pass, q * ds);
    }
    for (int nt = 2; nt < ntr_loc; nt++) {
      const Pc5 cm5 = nd_pc(A, np, cm, ks_m, nt), cp5 = nd_pc(A, np, cp, ks_p, nt);
      const double dtr = nd_pmeval(cm5, xm0, xm1) - nd_pmeval(cp5, xp0, xp1);
      const size_t otr = (size_t)(nt - 2) * 2 * kk * np;
      const bool pass = dtr * (V.f[F_trc][om + otr] - V.f[F_trc][op + otr]) >= 0. && dtr * (nd_tni(cm5, km0, xm0) - nd_tni(cp5, kp0, xp0)) >= 0. &&
                        dtr * (nd_tni(cm5, km1, xm1) - nd_tni(cp5, kp1, xp1)) >= 0.;
      rf[(size_t)nt * nf] = flux(pass, q * dtr);
    }
  }
}

// :886-906: the heat and salt flux of every neutral layer distributed over the layers of the velocity point it crosses
// (utflld.., utflx..); one thread per face walks its records, the layer index kuv only moves down
__global__ __launch_bounds__(64) void k_ndiff_uvflx(const DevView *__restrict__ Vp, NdArgs A) {
  const DevView &V = *Vp;
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t np = V.nplane, nf = 2 * np;
  if ((size_t)t_ >= nf) return;
  const size_t face = t_;
  const int n = A.rec_n[face];
  if (n == 0) return;
  const bool isv = face >= np;
  const size_t cp = isv ? face - np : face;
  const int kk = V.kk, mm = A.mm, ntr_loc = A.ntr_loc;
  const double *puv = A.puv + (isv ? (size_t)(kk + 1) * np : 0);
  gd_t ftl = (isv ? V.f[F_vtflld] : V.f[F_utflld]), fsl = (isv ? V.f[F_vsflld] : V.f[F_usflld]);
  gd_t ftx = (isv ? V.f[F_vtflx] : V.f[F_utflx]), fsx = (isv ? V.f[F_vsflx] : V.f[F_usflx]);
  int kuv = 1;
  // The sums of the layer kuv stay in registers until kuv moves on.  A move must not wait for memory: the lanes of a wave move at
  // different records, so nearly every record some lane moves, and a load in that path -- even one requested long before for the
  // lane that moves now -- costs the WAVE a full round trip (its load counter is shared).  So the column of interface pressures goes
  // to LDS first (each lane its own column: no barrier), and the sums of a layer start from zero without being read where they are
  // known to be zero: utflld .. always (k_ndiff_prep has just zeroed them), utflx .. inside blomgpu_step (flux_zero).
  HIP_DYNAMIC_SHARED(double, lds_uvf)
  double *pl = lds_uvf + threadIdx.x;
  for (int k0 = 0; k0 <= kk; k0 += ND_RB) {
    double a[ND_RB];
#pragma unroll
    for (int u = 0; u < ND_RB; u++) a[u] = puv[cp + (size_t)(k0 + u <= kk ? k0 + u : kk) * np];
#pragma unroll
    for (int u = 0; u < ND_RB; u++)
      if (k0 + u <= kk) pl[(k0 + u) * 64] = a[u];
  }
  const bool fz = A.flux_zero != 0;
  size_t ok = cp + (size_t)(kuv - 1 + mm) * np;
  double atl = 0., asl = 0., atx = fz ? 0. : ftx[ok], asx = fz ? 0. : fsx[ok], pk = pl[0], pk1 = pl[64];
  for (int r0 = 0; r0 < n; r0 += ND_RB) {
    double tf[ND_RB], sf[ND_RB], pup[ND_RB], plo[ND_RB];
#pragma unroll
    for (int u = 0; u < ND_RB; u++) {
      const int r = r0 + u < n ? r0 + u : n - 1;
      const double *rf = A.rec_f + face + (size_t)r * ntr_loc * nf;
      const double *rg = A.rec_g + face + (size_t)r * 7 * nf;
      tf[u] = rf[0]; sf[u] = rf[nf]; pup[u] = rg[5 * nf]; plo[u] = rg[6 * nf];
    }
#pragma unroll
    for (int u = 0; u < ND_RB; u++) {
      if (r0 + u >= n) break;
      const double tflx = tf[u], sflx = sf[u];
      if (!(tflx == tflx)) continue;
      const double p_ni_up = pup[u], p_ni_lo = plo[u];
      const double dp_ni_i = 1. / fmax2(ND_EPSILP, p_ni_lo - p_ni_up);
      while (kuv <= kk) {
        if (pk1 < p_ni_lo) {
          const double mlfrac = fmax2(0., pk1 - fmax2(p_ni_up, pk)) * dp_ni_i;
          atl = atl + tflx * mlfrac; asl = asl + sflx * mlfrac;
          atx = atx + tflx * mlfrac; asx = asx + sflx * mlfrac;
          ftl[ok] = atl; fsl[ok] = asl; ftx[ok] = atx; fsx[ok] = asx;
          kuv = kuv + 1;
          if (kuv <= kk) {
            ok = cp + (size_t)(kuv - 1 + mm) * np;
            atl = 0.; asl = 0.;
            if (fz) { atx = 0.; asx = 0.; } else { atx = ftx[ok]; asx = fsx[ok]; }
            pk = pk1; pk1 = pl[kuv * 64];
          }
        } else {
          const double mlfrac = (p_ni_lo - fmax2(p_ni_up, pk)) * dp_ni_i;
          atl = atl + tflx * mlfrac; asl = asl + sflx * mlfrac;
          atx = atx + tflx * mlfrac; asx = asx + sflx * mlfrac;
          break;
        }
      }
    }
  }
  if (kuv <= kk) { ftl[ok] = atl; fsl[ok] = asl; ftx[ok] = atx; fsx[ok] = asx; }
}

// the records of a cell's four faces replayed in the order of the j-slice loop (header): - v-face j, - u-face i, + u-face i+1,
// + v-face j+1.  One thread per cell AND field (blockIdx.y): the running sums of different fields do not meet.  Within a face
// the destination layer never decreases, so the sum of the current layer stays in a register until the layer changes.
__global__ __launch_bounds__(64) void k_ndiff_apply(const DevView *__restrict__ Vp, NdArgs A) {
  const DevView &V = *Vp;
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  const size_t c = t_, np = V.nplane, nf = 2 * np;
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int ntr_loc = A.ntr_loc, nt = blockIdx.y, kk = A.kk;
  const size_t faces[4] = {np + c, c, c + 1, np + c + V.ni};
  double *fl = A.flx + c + (size_t)nt * np;                      // layer stride ntr_loc * np
  // The cell's column of sums lives in LDS while the four faces are replayed (each lane its own column: no barrier) and goes to
  // memory once at the end: a change of layer must not read memory -- the lanes of a wave change layers at different records, and a
  // load in that path costs the whole wave a round trip every time (k_ndiff_uvflx).  k_ndiff_prep has zeroed flx: the sums start at 0.
  HIP_DYNAMIC_SHARED(double, lds_app)
  double *fc = lds_app + threadIdx.x;
  for (int k = 0; k < kk; k++) fc[k * 64] = 0.;
  for (int f = 0; f < 4; f++) {
    const size_t face = faces[f];
    const int n = A.rec_n[face];
    const bool plus = f < 2;                                     // this cell is the face's "plus" column: the flux leaves it
    int cur = 0;
    double acc = 0.;
    const int *rk = A.rec_k + face;
    const double *rf = A.rec_f + face + (size_t)nt * nf;
    for (int r0 = 0; r0 < n; r0 += ND_RB) {                      // the records of a batch are loaded together
      int kdds[ND_RB];
      double vs[ND_RB];
#pragma unroll
      for (int u = 0; u < ND_RB; u++) {
        const int r = r0 + u < n ? r0 + u : n - 1;
        kdds[u] = rk[(size_t)r * nf];
        vs[u] = rf[(size_t)r * ntr_loc * nf];
      }
#pragma unroll
      for (int u = 0; u < ND_RB; u++) {
        if (r0 + u >= n) break;
        const int kd = plus ? (kdds[u] >> 16) : (kdds[u] & 0xffff);
        const double v = vs[u];
        if (kd != cur) {
          if (cur) fc[(cur - 1) * 64] = acc;
          cur = kd;
          acc = fc[(cur - 1) * 64];
        }
        if (v == v) acc = plus ? acc - v : acc + v;
      }
    }
    if (cur) fc[(cur - 1) * 64] = acc;
  }
  for (int k = 0; k < kk; k++) fl[(size_t)k * ntr_loc * np] = fc[k * 64];
}

size_t ndiff_scratch_planes(int kk) { return (size_t)4 * kk + (size_t)10 * (kk + 1); }   // per FACE (2 nplane of them)

int st_ndiff_prep_flux(blomgpu_ctx *c, hipStream_t st, hipEvent_t ev_snap, NdArgs A, int *ksmx, int *kdmx, double *tsd, double *drt, double *drs) {
  const DevView &h = c->h;
  if (h.kk > 128) return ctx_fail(c, "ndiff: more than 128 layers");
  const unsigned nb = (unsigned)((h.nplane + 63) / 64);
  TimeScope ts(c, "ndiff", st);
  {
    TimeScope t1(c, "k_ndiff_prep", st);
    hipLaunchKernelGGL(k_ndiff_prep, dim3((unsigned)((h.nplane + 255) / 256)), dim3(256), 0, st, c->d, A, ksmx, kdmx, tsd, drt, drs);
  }
  if (ev_snap) HIPCHK(c, hipEventRecord(ev_snap, st));
  if (int rc = ctx_err_words(c)) return rc;
  {
    TimeScope t1(c, "k_ndiff_flux", st);
    if (h.kk <= 32) hipLaunchKernelGGL((k_ndiff_flux<1, 1>), dim3(nb, 2), dim3(64), 0, st, c->d, A, c->err_dev + 4);
    else if (h.kk <= 64) hipLaunchKernelGGL((k_ndiff_flux<1, 2>), dim3(nb, 2), dim3(64), 0, st, c->d, A, c->err_dev + 4);
    else hipLaunchKernelGGL((k_ndiff_flux<2, 4>), dim3(nb, 2), dim3(64), 0, st, c->d, A, c->err_dev + 4);
  }
  {
    TimeScope t1(c, "k_ndiff_eval", st);
    hipLaunchKernelGGL(k_ndiff_eval, dim3((unsigned)((2 * h.nplane + 63) / 64), ND_EVAL_RY), dim3(64), 0, st, c->d, A);
  }
  {
    TimeScope t1(c, "k_ndiff_uvflx", st);
    hipLaunchKernelGGL(k_ndiff_uvflx, dim3((unsigned)((2 * h.nplane + 63) / 64)), dim3(64), sizeof(double) * 64 * (h.kk + 1), st, c->d, A);
  }
  {
    TimeScope t1(c, "k_ndiff_apply", st);
    hipLaunchKernelGGL(k_ndiff_apply, dim3(nb, A.ntr_loc), dim3(64), sizeof(double) * 64 * h.kk, st, c->d, A);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
