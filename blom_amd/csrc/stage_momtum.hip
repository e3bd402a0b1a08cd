// momtum -- baroclinic momentum equation, phy/mod_momtum.F90:215-1282.
//
// The reference loops over layers (k-parallel OpenMP) and, per layer, over ~30 2-D sweeps with
// 2-D temporaries.  Layers are independent until the final vertical pass, so every sweep here is
// one kernel over all layers (blockIdx.y = layer), thread per point of the padded plane, the
// temporaries living in kk-level work-space fields:
//   k_mom_pupv, k_pscan        p,pu,pv from dp,dpu,dpv at the mid level               (:245-254,:322-338)
//   k_mom_drag                 bottom drag coefficient and friction velocity (column) (:260-292)
//   k_mom_tot                  utotm/n, vtotm/n, uflux, vflux, dpmx                    (:360-431)
//   k_mom_wall                 side-wall weights, uja/ujb/via/vib, del2 fields         (:438-472)
//   k_mom_vort                 vorticity/PV/deformation at boundary + interior points,
//                              kinetic energy                                         (:477-585,:613-629)
//   k_mom_visc                 deformation dependent viscosities at u- and v-points    (:829-841,:988-1000)
//   k_mom_flux1                longitudinal turbulent momentum fluxes at p-points      (:860-873,:1019-1034)
//   k_mom_update               lateral fluxes, Coriolis/advection, stresses, PGF time average and
//                              the update of u,v at both time levels                  (:723-813,:879-980,:1040-1143)
//   k_mom_column               massless-layer fill, barotropic extraction, time filter part 2,
//                              pu/pv at the new level (column)                        (:1154-1267)
// "first/last point of a wet segment" logic (ifu/ilu/jfu/jlu lists, phy/mod_bigrid.F90:320-429)
// is expressed with the masks: i is a segment start iff iu(i,j)=1 and iu(i-1,j)=0, etc.; where
// several sweeps of the reference write the same q-point the last writer in its order wins.
// Algorithmic bytes: 26 F plus the work-space round trips.  A fused LDS-tiled layer kernel
// (tile + 3-cell rim, temporaries in LDS, layer loop with coefficients resident) was built and verified
// bit-identical, and measured SLOWER on MI355X (2.7 ms vs 1.7 ms): the stage needs ~2000 fp64
// instructions per point (25 divisions), so the 2.1x rim recomputation makes it ALU bound (~0.7 ms
// at perfect issue) and the single 9-wave workgroup per CU cannot hide its own latencies across
// seven barriers per layer.  The sweeps therefore stay separate kernels.
// Roofline: HBM.
#include "blomgpu_internal.h"

#include "momtum_common.h"

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// work-space slots (kk-level fields)
enum {
  M_UTOTM, M_UFLUX, M_UTOTN, M_VTOTM, M_VFLUX, M_VTOTN, M_DPMX, M_WGTJA, M_WGTJB, M_UJA, M_UJB, M_DL2U,
  M_WGTIA, M_WGTIB, M_VIA, M_VIB, M_DL2V, M_POTVOR, M_DEFOR1, M_DEFOR2, M_KE, M_VSC2U, M_VSC4U, M_VSC2V,
  M_VSC4V, M_UFLUX1, M_VFLUX1, M_UHMIN, M_UHMAX, M_VHMIN, M_VHMAX, M_NSLOT
};
#define S2_DRAG 3      // 2-D work plane


// pu(k+1) = pu(k) + dpu(k+off), same for pv; range lo..+hi (:322-338 with lo=-1,hi=2; :1252-1267 interior)
__global__ void k_mom_pupv(const DevView *__restrict__ Vp, int off, int lo, int hi) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < lo || j > V.jj + hi || i < lo || i > V.ii + hi) return;
  const size_t np = V.nplane;
  if (V.m[I_iu][c]) column_scan(V.f[F_pu][c], V.f[F_dpu] + (size_t)off * np + c, V.f[F_pu] + c, np, V.kk);
  if (V.m[I_iv][c]) column_scan(V.f[F_pv][c], V.f[F_dpv] + (size_t)off * np + c, V.f[F_pv] + c, np, V.kk);
}

// p(k+1) = p(k) + dp(k+off) for j,i = lo..+hi
__global__ void k_mom_pscan(const DevView *__restrict__ Vp, int off, int lo, int hi) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < lo || j > V.jj + hi || i < lo || i > V.ii + hi || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  column_scan(V.f[F_p][c], V.f[F_dp] + (size_t)off * np + c, V.f[F_p] + c, np, V.kk);
}

// ---- :260-292 bottom drag ------------------------------------------------------------------------
__global__ void k_mom_drag(const DevView *__restrict__ Vp, int n, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj || i < 0 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, on = (size_t)(n - 1) * np, e = c + 1, nb = c + V.ni;
  const double thkbop = THKBOT * ONEM, tsfac = V.P.dlt / V.P.delt1;
  const double *p = V.f[F_p];
  const double pbot = p[c + (size_t)V.kk * np];
  double u1 = 0., u2 = 0.;
  for (int k = 0; k < V.kk; k++) {
    const size_t okn = (size_t)(k + nn) * np;
    const double pbotl = fmax2(p[c + (size_t)(k + 1) * np], pbot - thkbop);
    const double ptopl = fmax2(p[c + (size_t)k * np], pbot - thkbop);
    u1 = u1 + (V.f[F_u][c + okn] + V.f[F_u][e + okn]) * (pbotl - ptopl);
    u2 = u2 + (V.f[F_v][c + okn] + V.f[F_v][nb + okn]) * (pbotl - ptopl);
  }
  V.f[F_util1][c] = u1;
  V.f[F_util2][c] = u2;
  const double *ubf = V.f[F_ubflxs_p] + on, *vbf = V.f[F_vbflxs_p] + on, *pbu = V.f[F_pbu] + on, *pbv = V.f[F_pbv] + on;
  const double ubot = (ubf[c] / fmax2(EPSILPL, pbu[c] * V.f[F_scuy][c]) + ubf[e] / fmax2(EPSILPL, pbu[e] * V.f[F_scuy][e])) * tsfac +
                      u1 / thkbop;
  const double vbot = (vbf[c] / fmax2(EPSILPL, pbv[c] * V.f[F_scvx][c]) + vbf[nb] / fmax2(EPSILPL, pbv[nb] * V.f[F_scvx][nb])) * tsfac +
                      u2 / thkbop;
  const double ubbl = .5 * sqrt(ubot * ubot + vbot * vbot);
  const double q = V.P.cb * (ubbl + V.P.cbar);
  WK2(V, S2_DRAG)[c] = q * GRAV / (ALPHA0 * thkbop);
  V.f[F_ustarb][c] = sqrt(q * ubbl);
}

// ---- :360-431 total velocities, fluxes, dpmx ---------------------------------------------------------
__global__ void k_mom_tot(const DevView *__restrict__ Vp, int m, int n, int mm, int nn, int klo) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2) return;
  const int k = by_ + klo, ni = V.ni, ii = V.ii, jj = V.jj;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np, okn = (size_t)(k + nn) * np;
  const size_t om = (size_t)(m - 1) * np, on = (size_t)(n - 1) * np;
  const double tsfac = V.P.dlt / V.P.delt1, cutoff = ONEM;
  const int *iu = V.m[I_iu], *iv = V.m[I_iv];
  const double *dp = V.f[F_dp] + okm;
  if (j >= 0 && i >= 0) {                       // dpmx, j,i = 0..+2
    double d = 8. * cutoff;
    if (iu[c]) d = fmax2(d, dp[c] + dp[c - 1]);
    if (iu[c - ni]) d = fmax2(d, dp[c - ni] + dp[c - ni - 1]);           // u-point (i,j-1), j-1 = -1..jj+1
    if (iv[c]) d = fmax2(d, dp[c] + dp[c - ni]);
    if (iv[c - 1]) d = fmax2(d, dp[c - 1] + dp[c - 1 - ni]);             // v-point (i-1,j), i-1 = -1..ii+1
    WK(V, M_DPMX)[c + ok] = d;
  }
  // Points without a u/v point hold 0 in the reference's (module) arrays (inivar_utility,
  // phy/mod_utility.F90:86-150); our work-space planes are shared between stages, so write the 0s.
  const bool in01 = j >= 0 && j <= jj + 1 && i >= 0 && i <= ii + 1;
  if (iu[c]) {
    const double scuy = V.f[F_scuy][c];
    if (in01) {
      const double ut = V.f[F_u][c + okm] + V.f[F_ubflxs_p][c + om] * tsfac / (V.f[F_pbu][c + om] * scuy);
      WK(V, M_UTOTM)[c + ok] = ut;
      WK(V, M_UFLUX)[c + ok] = ut * fmax2(V.f[F_dpu][c + okm], cutoff);
    }
    const double un = V.f[F_u][c + okn] + V.f[F_ubflxs_p][c + on] * tsfac / (V.f[F_pbu][c + on] * scuy);
    WK(V, M_UTOTN)[c + ok] = un;
    // the reference's module array utotn is left holding the last layer's values outside the
    // interior (the interior receives the barotropic forcing at :1158-1175, :1238)
    if (k == V.kk - 1) V.f[F_utotn][c] = un;
  } else {
    if (in01) { WK(V, M_UTOTM)[c + ok] = 0.; WK(V, M_UFLUX)[c + ok] = 0.; }
    WK(V, M_UTOTN)[c + ok] = 0.;
  }
  if (iv[c]) {
    const double scvx = V.f[F_scvx][c];
    if (in01) {
      const double vt = V.f[F_v][c + okm] + V.f[F_vbflxs_p][c + om] * tsfac / (V.f[F_pbv][c + om] * scvx);
      WK(V, M_VTOTM)[c + ok] = vt;
      WK(V, M_VFLUX)[c + ok] = vt * fmax2(V.f[F_dpv][c + okm], cutoff);
    }
    const double vn = V.f[F_v][c + okn] + V.f[F_vbflxs_p][c + on] * tsfac / (V.f[F_pbv][c + on] * scvx);
    WK(V, M_VTOTN)[c + ok] = vn;
    if (k == V.kk - 1) V.f[F_vtotn][c] = vn;
  } else {
    if (in01) { WK(V, M_VTOTM)[c + ok] = 0.; WK(V, M_VFLUX)[c + ok] = 0.; }
    WK(V, M_VTOTN)[c + ok] = 0.;
  }
}

// ---- :438-472 side-wall weights, auxiliary velocities, del2 fields -------------------------------------
__global__ void k_mom_wall(const DevView *__restrict__ Vp, int m, int klo) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const int k = by_ + klo, ni = V.ni;
  const size_t np = V.nplane, ok = (size_t)k * np, om = (size_t)(m - 1) * np;
  if (V.m[I_iu][c] && j >= -1 && j <= V.jj + 2 && i >= 0 && i <= V.ii + 2) {
    const double *utotn = WK(V, M_UTOTN) + ok;
    const double pu1 = V.f[F_pu][c + (size_t)(k + 1) * np], pu0 = V.f[F_pu][c + ok];
    const double den = fmax2(pu1 - pu0, EPSILP);
    const double wa = fmax2(0., fmin2(1., (pu1 - V.f[F_pbu][c - ni + om]) / den));
    const double wb = fmax2(0., fmin2(1., (pu1 - V.f[F_pbu][c + ni + om]) / den));
    const double un = utotn[c];
    const double uja = (1. - wa) * utotn[c - ni] + wa * SLIP * un;
    const double ujb = (1. - wb) * utotn[c + ni] + wb * SLIP * un;
    WK(V, M_WGTJA)[c + ok] = wa;
    WK(V, M_WGTJB)[c + ok] = wb;
    WK(V, M_UJA)[c + ok] = uja;
    WK(V, M_UJB)[c + ok] = ujb;
    WK(V, M_DL2U)[c + ok] = un - .25 * (utotn[c + 1] + utotn[c - 1] + uja + ujb);
  } else if (j >= -1 && j <= V.jj + 2 && i >= 0 && i <= V.ii + 2) {
    WK(V, M_UJA)[c + ok] = 0.; WK(V, M_UJB)[c + ok] = 0.; WK(V, M_DL2U)[c + ok] = 0.;   // cf. inivar_momtum :177-189
  }
  if (V.m[I_iv][c] && j >= 0 && j <= V.jj + 2 && i >= -1 && i <= V.ii + 2) {
    const double *vtotn = WK(V, M_VTOTN) + ok;
    const double pv1 = V.f[F_pv][c + (size_t)(k + 1) * np], pv0 = V.f[F_pv][c + ok];
    const double den = fmax2(pv1 - pv0, EPSILP);
    const double wa = fmax2(0., fmin2(1., (pv1 - V.f[F_pbv][c - 1 + om]) / den));
    const double wb = fmax2(0., fmin2(1., (pv1 - V.f[F_pbv][c + 1 + om]) / den));
    const double vn = vtotn[c];
    const double via = (1. - wa) * vtotn[c - 1] + wa * SLIP * vn;
    const double vib = (1. - wb) * vtotn[c + 1] + wb * SLIP * vn;
    WK(V, M_WGTIA)[c + ok] = wa;
    WK(V, M_WGTIB)[c + ok] = wb;
    WK(V, M_VIA)[c + ok] = via;
    WK(V, M_VIB)[c + ok] = vib;
    WK(V, M_DL2V)[c + ok] = vn - .25 * (vtotn[c + ni] + vtotn[c - ni] + via + vib);
  } else if (j >= 0 && j <= V.jj + 2 && i >= -1 && i <= V.ii + 2) {
    WK(V, M_VIA)[c + ok] = 0.; WK(V, M_VIB)[c + ok] = 0.; WK(V, M_DL2V)[c + ok] = 0.;
  }
}

// ---- :477-585 vorticity, potential vorticity, deformation; :613-629 kinetic energy ---------------------
__global__ void k_mom_vort(const DevView *__restrict__ Vp, int mm, int klo) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2) return;
  const int k = by_ + klo, ni = V.ni, ii = V.ii, jj = V.jj;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np;
  const int *ip = V.m[I_ip], *iu = V.m[I_iu], *iv = V.m[I_iv], *iq = V.m[I_iq];
  const double *utotm = WK(V, M_UTOTM) + ok, *vtotm = WK(V, M_VTOTM) + ok;
  const double *utotn = WK(V, M_UTOTN) + ok, *vtotn = WK(V, M_VTOTN) + ok;
  const double *dpmx = WK(V, M_DPMX) + ok, *dp = V.f[F_dp] + okm;
  const double *scvy = V.f[F_scvy], *scux = V.f[F_scux], *scq2i = V.f[F_scq2i];
  const size_t w = c - 1, s = c - ni, sw = c - 1 - ni;

  // --- vorticity / PV at q-point (i,j), j,i = 1..+1; last writer in the reference's sweep order wins
  if (j >= 1 && j <= jj + 1 && i >= 1 && i <= ii + 1) {
    bool have = false;
    double vort = 0., dpv = 1.;
    if (iv[c] && !iv[w]) {                           // first point of a v-segment, :479-486
      vort = vtotm[c] * (1. - SLIP) * scvy[c] * scq2i[c];
      dpv = .125 * fmax2(fmax2(4. * (dp[c] + dp[s]), dpmx[c]), dpmx[c + 1]);
      have = true;
    } else if (iv[w] && !iv[c]) {                    // one past the last point (i-1) of a v-segment, :487-494
      vort = -vtotm[w] * (1. - SLIP) * scvy[w] * scq2i[c];
      dpv = .125 * fmax2(fmax2(4. * (dp[w] + dp[sw]), dpmx[w]), dpmx[c]);
      have = true;
    }
    if (iu[c] && !iu[s]) {                           // first point (in j) of a u-segment, :513-520
      vort = -utotm[c] * (1. - SLIP) * scux[c] * scq2i[c];
      dpv = .125 * fmax2(fmax2(4. * (dp[c] + dp[w]), dpmx[c]), dpmx[c + ni]);
      have = true;
    } else if (iu[s] && !iu[c]) {                    // one past the last point (j-1), :521-528
      vort = utotm[s] * (1. - SLIP) * scux[s] * scq2i[c];
      dpv = .125 * fmax2(fmax2(4. * (dp[s] + dp[sw]), dpmx[s]), dpmx[c]);
      have = true;
    }
    if (iq[c]) {                                     // interior (incl. promontories), :561-575
      vort = (vtotm[c] * scvy[c] - vtotm[w] * scvy[w] - utotm[c] * scux[c] + utotm[s] * scux[s]) * scq2i[c];
      double d = fmax2(2. * (dp[c] + dp[w] + dp[s] + dp[sw]), dpmx[c]);
      d = fmax2(d, dpmx[w]);
      d = fmax2(d, dpmx[c + 1]);
      d = fmax2(d, dpmx[s]);
      d = fmax2(d, dpmx[c + ni]);
      dpv = .125 * d;
      have = true;
    }
    if (have) {
      const double av = vort + V.f[F_corioq][c];
      V.f[F_absvor][c + ok] = av;
      V.f[F_dpvor][c + ok] = dpv;
      WK(V, M_POTVOR)[c + ok] = av / dpv;
    }
  }
  // --- defor2 at q-point (i,j), j,i = 0..+2
  if (j >= 0 && i >= 0) {
    bool have = false;
    double d2 = 0.;
    if (iv[c] && !iv[w]) { const double t = vtotn[c] * (1. - SLIP) * scvy[c]; d2 = t * t * scq2i[c]; have = true; }          // :500-503
    else if (iv[w] && !iv[c]) { const double t = vtotn[w] * (1. - SLIP) * scvy[w]; d2 = t * t * scq2i[c]; have = true; }     // :504-507
    if (iu[c] && !iu[s]) { const double t = utotn[c] * (1. - SLIP) * scux[c]; d2 = t * t * scq2i[c]; have = true; }          // :534-537
    else if (iu[s] && !iu[c]) { const double t = utotn[s] * (1. - SLIP) * scux[s]; d2 = t * t * scq2i[c]; have = true; }     // :538-541
    if (iq[c]) {                                                                                                             // :577-585
      const double t = WK(V, M_VIB)[w + ok] * scvy[c] - WK(V, M_VIA)[c + ok] * scvy[w] + WK(V, M_UJB)[s + ok] * scux[c] -
                       WK(V, M_UJA)[c + ok] * scux[s];
      d2 = t * t * scq2i[c];
      have = true;
    }
    if (have) WK(V, M_DEFOR2)[c + ok] = d2;
  }
  if (!ip[c]) return;
  // --- defor1 at p-points, j,i = -1..+1 (:549-559)
  if (j <= jj + 1 && i <= ii + 1) {
    const double t = (utotn[c + 1] * V.f[F_scuy][c + 1] - utotn[c] * V.f[F_scuy][c]) -
                     (vtotn[c + ni] * V.f[F_scvx][c + ni] - vtotn[c] * V.f[F_scvx][c]);
    WK(V, M_DEFOR1)[c + ok] = t * t * V.f[F_scp2i][c];
  }
  // --- kinetic energy (GOLD version of Arakawa-Lamb), j,i = 0..jj/ii (:613-629)
  if (j >= 0 && j <= jj && i >= 0 && i <= ii) {
    const double ue = utotm[c + 1], uw = utotm[c], vn = vtotm[c + ni], vs = vtotm[c];
    WK(V, M_KE)[c + ok] = .25 * (V.f[F_scu2][c] * (uw * uw) + V.f[F_scu2][c + 1] * (ue * ue) + V.f[F_scv2][c] * (vs * vs) +
                                 V.f[F_scv2][c + ni] * (vn * vn)) / V.f[F_scp2][c];
  }
}

__global__ void k_mom_enedis(const DevView *__restrict__ Vp, int mm, int klo) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1) return;
  const int k = by_ + klo;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np;
  const double *dp = V.f[F_dp] + okm;
  // the reference's local arrays are zeroed once per call (:238-241) and written at wet points only
  double a = 0., b = 0.;
  if (V.m[I_iu][c]) enedis_minmax(.5 * WK(V, M_UTOTM)[c + ok] * (dp[c] + dp[c - 1]), WK(V, M_UFLUX)[c + ok], a, b);
  WK(V, M_UHMIN)[c + ok] = a; WK(V, M_UHMAX)[c + ok] = b;
  a = 0.; b = 0.;
  if (V.m[I_iv][c]) enedis_minmax(.5 * WK(V, M_VTOTM)[c + ok] * (dp[c] + dp[c - V.ni]), WK(V, M_VFLUX)[c + ok], a, b);
  WK(V, M_VHMIN)[c + ok] = a; WK(V, M_VHMAX)[c + ok] = b;
}

// ---- :829-841 and :988-1000 deformation dependent viscosities ---------------------------------------
__global__ void k_mom_visc(const DevView *__restrict__ Vp, int klo) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1) return;
  const int k = by_ + klo, ni = V.ni;
  const size_t ok = (size_t)k * V.nplane;
  const double *d1 = WK(V, M_DEFOR1) + ok, *d2 = WK(V, M_DEFOR2) + ok, *difwgt = V.f[F_difwgt];
  const Params &P = V.P;
  if (V.m[I_iu][c]) {
    const double q = .5 * (difwgt[c - 1] + difwgt[c]);
    const double deform = sqrt(.5 * (d1[c] + d1[c - 1] + d2[c] + d2[c + ni]));
    WK(V, M_VSC2U)[c + ok] = fmax2(q * P.mdv2hi + (1. - q) * P.mdv2lo, (q * P.vsc2hi + (1. - q) * P.vsc2lo) * deform);
    WK(V, M_VSC4U)[c + ok] = fmax2(q * P.mdv4hi + (1. - q) * P.mdv4lo, (q * P.vsc4hi + (1. - q) * P.vsc4lo) * deform);
  }
  if (V.m[I_iv][c]) {
    const double q = .5 * (difwgt[c - ni] + difwgt[c]);
    const double deform = sqrt(.5 * (d1[c] + d1[c - ni] + d2[c] + d2[c + 1]));
    WK(V, M_VSC2V)[c + ok] = fmax2(q * P.mdv2hi + (1. - q) * P.mdv2lo, (q * P.vsc2hi + (1. - q) * P.vsc2lo) * deform);
    WK(V, M_VSC4V)[c + ok] = fmax2(q * P.mdv4hi + (1. - q) * P.mdv4lo, (q * P.vsc4hi + (1. - q) * P.vsc4lo) * deform);
  }
}

// ---- :860-873 and :1019-1034 longitudinal turbulent momentum fluxes at p-points ------------------------
__global__ void k_mom_flux1(const DevView *__restrict__ Vp, int mm, int klo) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj || i < 0 || i > V.ii || !V.m[I_ip][c]) return;
  const int k = by_ + klo, ni = V.ni;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np;
  const int *iu = V.m[I_iu], *iv = V.m[I_iv];
  const double difmxp = V.f[F_difmxp][c];
  if (j >= 1 && iu[c] + iu[c + 1] > 0) {
    const double *dpu = V.f[F_dpu] + okm, *utotn = WK(V, M_UTOTN) + ok, *dl2u = WK(V, M_DL2U) + ok;
    const double dpxy = fmax2(dpu[c], ONEMM), dpib = fmax2(dpu[c + 1], ONEMM);
    const double v2 = ext_i(iu, WK(V, M_VSC2U) + ok, c) + ext_i(iu, WK(V, M_VSC2U) + ok, c + 1);
    const double v4 = ext_i(iu, WK(V, M_VSC4U) + ok, c) + ext_i(iu, WK(V, M_VSC4U) + ok, c + 1);
    WK(V, M_UFLUX1)[c + ok] = fmin2(difmxp, v2 * V.f[F_scpy][c]) * hfharm(dpxy, dpib) * (utotn[c] - utotn[c + 1]) +
                              fmin2(.125 * difmxp, v4 * V.f[F_scpy][c]) * hfharm(dpxy, dpib) * (dl2u[c] - dl2u[c + 1]);
  }
  if (i >= 1 && iv[c] + iv[c + ni] > 0) {
    const double *dpv = V.f[F_dpv] + okm, *vtotn = WK(V, M_VTOTN) + ok, *dl2v = WK(V, M_DL2V) + ok;
    const double dpxy = fmax2(dpv[c], ONEMM), dpjb = fmax2(dpv[c + ni], ONEMM);
    const double v2 = ext_j(iv, WK(V, M_VSC2V) + ok, c, ni) + ext_j(iv, WK(V, M_VSC2V) + ok, c + ni, ni);
    const double v4 = ext_j(iv, WK(V, M_VSC4V) + ok, c, ni) + ext_j(iv, WK(V, M_VSC4V) + ok, c + ni, ni);
    WK(V, M_VFLUX1)[c + ok] = fmin2(difmxp, v2 * V.f[F_scpx][c]) * hfharm(dpxy, dpjb) * (vtotn[c] - vtotn[c + ni]) +
                              fmin2(.125 * difmxp, v4 * V.f[F_scpx][c]) * hfharm(dpxy, dpjb) * (dl2v[c] - dl2v[c + ni]);
  }
}

// ---- update of u and v at interior points: :723-813, :879-980, :1040-1143 --------------------------------
__global__ void k_mom_update(const DevView *__restrict__ Vp, int m, int mm, int nn, int klo) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const int k = by_ + klo, ni = V.ni;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np, okn = (size_t)(k + nn) * np;
  const size_t om = (size_t)(m - 1) * np;
  const int *iu = V.m[I_iu], *iv = V.m[I_iv];
  const Params &P = V.P;
  const double delt1 = P.delt1, tsfac = P.dlt / P.delt1, thkbop = THKBOT * ONEM;
  const double *potvor = WK(V, M_POTVOR) + ok, *ke = WK(V, M_KE) + ok, *drag = WK2(V, S2_DRAG);
  const double *p0 = V.f[F_p] + ok, *p1 = V.f[F_p] + (size_t)(k + 1) * np;
  if (iu[c]) {
    const double *vflux = WK(V, M_VFLUX) + ok, *utotn = WK(V, M_UTOTN) + ok, *dl2u = WK(V, M_DL2U) + ok;
    const double *dpu = V.f[F_dpu] + okm;
    const size_t w = c - 1, s = c - ni, nb = c + ni;
    double cau;
    if (P.mommth == 0)
      cau = .125 * (vflux[c] + vflux[nb] + vflux[w] + vflux[w + ni]) * (potvor[c] + potvor[nb]);
    else if (P.mommth == 1)
      cau = .25 * ((vflux[c] + vflux[w]) * potvor[c] + (vflux[nb] + vflux[w + ni]) * potvor[nb]);
    else {                                                                   // enedis, :771-790
      const double *vhmx = WK(V, M_VHMAX) + ok, *vhmn = WK(V, M_VHMIN) + ok;
      const double utm = WK(V, M_UTOTM)[c + ok];
      double t1, t2;
      const double pn = potvor[nb], pc = potvor[c];
      if (pn * utm == 0.) t1 = pn * ((vhmx[nb] + vhmx[w + ni]) + (vhmn[nb] + vhmn[w + ni])) * .5;
      else if (pn * utm < 0.) t1 = pn * (vhmx[nb] + vhmx[w + ni]);
      else t1 = pn * (vhmn[nb] + vhmn[w + ni]);
      if (pc * utm == 0.) t2 = pc * ((vhmx[c] + vhmx[w]) + (vhmn[c] + vhmn[w])) * .5;
      else if (pc * utm < 0.) t2 = pc * (vhmx[c] + vhmx[w]);
      else t2 = pc * (vhmn[c] + vhmn[w]);
      cau = .25 * (t1 + t2);
    }
    // lateral turbulent momentum fluxes, :879-913
    const double wja = WK(V, M_WGTJA)[c + ok], wjb = WK(V, M_WGTJB)[c + ok];
    const double dpxy = fmax2(dpu[c], ONEMM);
    double dpja = fmax2(dpu[s], ONEMM);
    dpja = dpja + wja * (dpxy - dpja);
    double dpjb = fmax2(dpu[nb], ONEMM);
    dpjb = dpjb + wjb * (dpxy - dpjb);
    const double *vsc2 = WK(V, M_VSC2U) + ok, *vsc4 = WK(V, M_VSC4U) + ok;
    const double vsc2a = iu[s] == 0 ? vsc2[c] : vsc2[s], vsc4a = iu[s] == 0 ? vsc4[c] : vsc4[s];
    const double vsc2b = iu[nb] == 0 ? vsc2[c] : vsc2[nb], vsc4b = iu[nb] == 0 ? vsc4[c] : vsc4[nb];
    const double un = utotn[c], d2 = dl2u[c];
    const double dl2uja = (1. - wja) * dl2u[s] + wja * SLIP * d2;          // :594-597
    const double dl2ujb = (1. - wjb) * dl2u[nb] + wjb * SLIP * d2;
    const double dmq0 = V.f[F_difmxq][c], dmq1 = V.f[F_difmxq][nb];
    const double uflux2 = fmin2(dmq0, (vsc2[c] + vsc2a) * V.f[F_scqx][c]) * hfharm(dpja, dpxy) * (WK(V, M_UJA)[c + ok] - un) +
                          fmin2(.125 * dmq0, (vsc4[c] + vsc4a) * V.f[F_scqx][c]) * hfharm(dpja, dpxy) * (dl2uja - d2);
    const double uflux3 = fmin2(dmq1, (vsc2[c] + vsc2b) * V.f[F_scqx][nb]) * hfharm(dpjb, dpxy) * (un - WK(V, M_UJB)[c + ok]) +
                          fmin2(.125 * dmq1, (vsc4[c] + vsc4b) * V.f[F_scqx][nb]) * hfharm(dpjb, dpxy) * (d2 - dl2ujb);
    // wind stress (isopyc_bulkml: top layer only), :919-936
    double stress = 0.;
    if (k == 0) stress = -2. * V.f[F_taux][c] * GRAV * V.f[F_scux][c] / (V.f[F_p][c + np] + V.f[F_p][w + np]);
    const double pbu = V.f[F_pbu][c + om];
    const double ptopl = .5 * (fmin2(pbu, p0[c]) + fmin2(pbu, p0[w]));
    const double pbotl = .5 * (fmin2(pbu, p1[c]) + fmin2(pbu, p1[w]));
    const double q = .5 * (drag[c] + drag[w]) * (fmax2(pbu - thkbop, pbotl) - fmax2(pbu - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                     fmax2(dpu[c], ONEMM);
    const double botstr = -un * q / (1. + delt1 * q);
    const double pgf = (1. - 2. * WPGF) * V.f[F_pgfx][c + okm] + WPGF * (V.f[F_pgfx_o][c + ok] + V.f[F_pgfx][c + okn]);
    const double ukm = V.f[F_u][c + okm], ukn = V.f[F_u][c + okn];
    V.f[F_u][c + okm] = ukm * (P.wuv1 * dpu[c] + ONEMM) + ukn * P.wuv2 * V.f[F_dpuold][c + ok];
    const double ubrhs = V.f[F_ubcors_p][c] * tsfac;                       // :302
    const double *uflux1 = WK(V, M_UFLUX1) + ok;
    V.f[F_u][c + okn] = ukn + delt1 * (-V.f[F_scuxi][c] * (-pgf + stress + (ke[c] - ke[w])) + cau - ubrhs + botstr -
                                       (uflux1[c] - uflux1[w] + uflux3 - uflux2) / (V.f[F_scu2][c] * fmax2(dpu[c], ONEMM)));
  }
  if (iv[c]) {
    const double *uflux = WK(V, M_UFLUX) + ok, *vtotn = WK(V, M_VTOTN) + ok, *dl2v = WK(V, M_DL2V) + ok;
    const double *dpv = V.f[F_dpv] + okm;
    const size_t w = c - 1, e = c + 1, s = c - ni;
    double cav;
    if (P.mommth == 0)
      cav = -.125 * (uflux[c] + uflux[e] + uflux[s] + uflux[e - ni]) * (potvor[c] + potvor[e]);
    else if (P.mommth == 1)
      cav = -.25 * ((uflux[c] + uflux[s]) * potvor[c] + (uflux[e] + uflux[e - ni]) * potvor[e]);
    else {                                                                   // enedis, :793-812
      const double *uhmx = WK(V, M_UHMAX) + ok, *uhmn = WK(V, M_UHMIN) + ok;
      const double vtm = WK(V, M_VTOTM)[c + ok];
      double t1, t2;
      const double pe = potvor[e], pc = potvor[c];
      if (pe * vtm == 0.) t1 = pe * ((uhmx[e] + uhmx[e - ni]) + (uhmn[e] + uhmn[e - ni])) * .5;
      else if (pe * vtm > 0.) t1 = pe * (uhmx[e] + uhmx[e - ni]);
      else t1 = pe * (uhmn[e] + uhmn[e - ni]);
      if (pc * vtm == 0.) t2 = pc * ((uhmx[c] + uhmx[s]) + (uhmn[c] + uhmn[s])) * .5;
      else if (pc * vtm > 0.) t2 = pc * (uhmx[c] + uhmx[s]);
      else t2 = pc * (uhmn[c] + uhmn[s]);
      cav = -.25 * (t1 + t2);
    }
    const double wia = WK(V, M_WGTIA)[c + ok], wib = WK(V, M_WGTIB)[c + ok];
    const double dpxy = fmax2(dpv[c], ONEMM);
    double dpia = fmax2(dpv[w], ONEMM);
    dpia = dpia + wia * (dpxy - dpia);
    double dpib = fmax2(dpv[e], ONEMM);
    dpib = dpib + wib * (dpxy - dpib);
    const double *vsc2 = WK(V, M_VSC2V) + ok, *vsc4 = WK(V, M_VSC4V) + ok;
    const double vsc2a = iv[w] == 0 ? vsc2[c] : vsc2[w], vsc4a = iv[w] == 0 ? vsc4[c] : vsc4[w];
    const double vsc2b = iv[e] == 0 ? vsc2[c] : vsc2[e], vsc4b = iv[e] == 0 ? vsc4[c] : vsc4[e];
    const double vn = vtotn[c], d2 = dl2v[c];
    const double dl2via = (1. - wia) * dl2v[w] + wia * SLIP * d2;          // :602-605
    const double dl2vib = (1. - wib) * dl2v[e] + wib * SLIP * d2;
    const double dmq0 = V.f[F_difmxq][c], dmq1 = V.f[F_difmxq][e];
    const double vflux2 = fmin2(dmq0, (vsc2[c] + vsc2a) * V.f[F_scqy][c]) * hfharm(dpia, dpxy) * (WK(V, M_VIA)[c + ok] - vn) +
                          fmin2(.125 * dmq0, (vsc4[c] + vsc4a) * V.f[F_scqy][c]) * hfharm(dpia, dpxy) * (dl2via - d2);
    const double vflux3 = fmin2(dmq1, (vsc2[c] + vsc2b) * V.f[F_scqy][e]) * hfharm(dpib, dpxy) * (vn - WK(V, M_VIB)[c + ok]) +
                          fmin2(.125 * dmq1, (vsc4[c] + vsc4b) * V.f[F_scqy][e]) * hfharm(dpib, dpxy) * (d2 - dl2vib);
    double stress = 0.;
    if (k == 0) stress = -2. * V.f[F_tauy][c] * GRAV * V.f[F_scvy][c] / (V.f[F_p][c + np] + V.f[F_p][s + np]);
    const double pbv = V.f[F_pbv][c + om];
    const double ptopl = .5 * (fmin2(pbv, p0[c]) + fmin2(pbv, p0[s]));
    const double pbotl = .5 * (fmin2(pbv, p1[c]) + fmin2(pbv, p1[s]));
    const double q = .5 * (drag[c] + drag[s]) * (fmax2(pbv - thkbop, pbotl) - fmax2(pbv - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                     fmax2(dpv[c], ONEMM);
    const double botstr = -vn * q / (1. + delt1 * q);
    const double pgf = (1. - 2. * WPGF) * V.f[F_pgfy][c + okm] + WPGF * (V.f[F_pgfy_o][c + ok] + V.f[F_pgfy][c + okn]);
    const double vkm = V.f[F_v][c + okm], vkn = V.f[F_v][c + okn];
    V.f[F_v][c + okm] = vkm * (P.wuv1 * dpv[c] + ONEMM) + vkn * P.wuv2 * V.f[F_dpvold][c + ok];
    const double vbrhs = V.f[F_vbcors_p][c] * tsfac;                       // :307
    const double *vflux1 = WK(V, M_VFLUX1) + ok;
    V.f[F_v][c + okn] = vkn + delt1 * (-V.f[F_scvyi][c] * (-pgf + stress + (ke[c] - ke[s])) + cav - vbrhs + botstr -
                                       (vflux1[c] - vflux1[s] + vflux3 - vflux2) / (V.f[F_scv2][c] * fmax2(dpv[c], ONEMM)));
  }
}

// ---- :1154-1267 vertical pass: massless layers, barotropic part, time filter part 2, pu/pv --------------
__global__ void k_mom_column(const DevView *__restrict__ Vp, int m, int mm, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  double *u = isv ? V.f[F_v] : V.f[F_u];
  const double *dpu = isv ? V.f[F_dpv] : V.f[F_dpu], *dpuold = isv ? V.f[F_dpvold] : V.f[F_dpuold];
  const double umax = (isv ? V.f[F_vmax] : V.f[F_umax])[c];
  const double ub = (isv ? V.f[F_vb] : V.f[F_ub])[c + (size_t)(m - 1) * np];
  const double wuv1 = V.P.wuv1, wuv2 = V.P.wuv2;
  double tot = 0., uabove = 0.;
  for (int k = 0; k < kk; k++) {
    const size_t okm = c + (size_t)(k + mm) * np, okn = c + (size_t)(k + nn) * np;
    const double dn = dpu[okn];
    const double q = fmin2(fmin2(dpu[okm], dn), ONEM);
    double un = u[okn];
    const double ukan = k == 0 ? un : uabove;                              // kan = max(1,k-1)+nn
    un = (un * q + ukan * (ONEM - q)) / ONEM;
    un = fmax2(-umax, fmin2(umax, un + ub)) - ub;
    u[okn] = un;
    uabove = un;
    tot = tot + un * dn;
  }
  tot = tot / (isv ? V.f[F_pbv_p] : V.f[F_pbu_p])[c];
  double pacc = (isv ? V.f[F_pv] : V.f[F_pu])[c];
  for (int k = 0; k < kk; k++) {
    const size_t okm = c + (size_t)(k + mm) * np, okn = c + (size_t)(k + nn) * np;
    const double dn = dpu[okn];
    const double un = u[okn] - tot;
    u[okn] = un;
    u[okm] = (u[okm] + un * wuv2 * dn) / (wuv1 * dpu[okm] + ONEMM + wuv2 * (dpuold[c + (size_t)k * np] + dn));
    pacc = pacc + dn;
    (isv ? V.f[F_pv] : V.f[F_pu])[c + (size_t)(k + 1) * np] = pacc;
  }
  (isv ? V.f[F_vtotn] : V.f[F_utotn])[c] = tot * (1. / V.P.delt1);
}

int st_momtum(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag != 1) return ctx_fail(c, "momtum: hybrid-coordinate wind stress (mu_nonloc) is not built yet");
  if (h.nwk < M_NSLOT) return ctx_fail(c, "momtum: device work space too small");
  const dim3 gcol = plane_grid(h, 1, 64), b(256), b64(64);
  TimeScope ts(c, "momtum");
  hipLaunchKernelGGL(k_mom_pscan, gcol, b64, 0, c->stream, c->d, mm, -1, 2);
  hipLaunchKernelGGL(k_mom_drag, gcol, b64, 0, c->stream, c->d, n, nn);
  hipLaunchKernelGGL(k_mom_pupv, gcol, b64, 0, c->stream, c->d, mm, -1, 2);
  if (int rc = st_xctilr(c, h.f[F_difwgt], 1, 1, 2, 2, 1)) return rc;                       // :340
  if (c->momtum_v == 2) return st_momtum_fused_layers(c, m, n, mm, nn);                     // stage_momtum_fused.hip
  // The layer loop (:342) in chunks: a chunk's ~55 planes (inputs, ~30 temporaries, outputs) of `ch` layers each are
  // produced and consumed by consecutive kernels while they still sit in the 256 MiB Infinity Cache.
  const int ch = c->momtum_chunk > 0 ? c->momtum_chunk : h.kk;
  for (int klo = 0; klo < h.kk; klo += ch) {
    const dim3 g = plane_grid(h, h.kk - klo < ch ? h.kk - klo : ch);
    hipLaunchKernelGGL(k_mom_tot, g, b, 0, c->stream, c->d, m, n, mm, nn, klo);
    hipLaunchKernelGGL(k_mom_wall, g, b, 0, c->stream, c->d, m, klo);
    hipLaunchKernelGGL(k_mom_vort, g, b, 0, c->stream, c->d, mm, klo);
    if (h.P.mommth == 2) hipLaunchKernelGGL(k_mom_enedis, g, b, 0, c->stream, c->d, mm, klo);
    hipLaunchKernelGGL(k_mom_visc, g, b, 0, c->stream, c->d, klo);
    hipLaunchKernelGGL(k_mom_flux1, g, b, 0, c->stream, c->d, mm, klo);
    hipLaunchKernelGGL(k_mom_update, g, b, 0, c->stream, c->d, m, mm, nn, klo);
  }
  hipLaunchKernelGGL(k_mom_column, plane_grid(h, 2, 64), b64, 0, c->stream, c->d, m, mm, nn);
  HIPCHK(c, hipGetLastError());
  return 0;
}
