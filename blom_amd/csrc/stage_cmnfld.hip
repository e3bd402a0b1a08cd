// cmnfld2 for vcoord = isopyc_bulkml -- phy/mod_cmnfld_routines.F90:1158-1238: the halo updates of temp, saln and
// kfpla (:1171-1196), the buoyancy frequency squared on interfaces, in layers and vertically filtered
// (cmnfld_bfsqf_isopyc_bulkml, :61-227), the interface geopotential and the slope of the local neutral surface at
// velocity points (cmnfld_nslope_isopyc_bulkml, :423-652) -- the producer of nslpx/nslpy that eddtra consumes.
// PARITY UNPINNED: mod_cmnfld_routines uses mod_dia (netCDF) and cannot be built in this image; the kernels agree bit
// for bit with the C restatement oracle/c/cmnfld.c written from the same lines, and both are checked by construction
// (tests/test_cmnfld.py).
// All three kernels are column kernels (one thread per column, coalesced plane by plane); HBM-bound, ~16 F.
#include "blomgpu_internal.h"
#include "eos.h"

#define GRAV 9.806
#define RHO0 1.e3
#define EPSILP 1.e-12
#define ONEM 9806.
#define ONEMM 9.806
// phy/mod_cmnfld.F90:36-46
#define SLS0 (10. * ONEM)
#define SLSMFQ 2.
#define SLSELS 2.
#define BFSQMN 1.e-7

// (CM_GAM behind the nine slots of difest_isobml's kernels, stage_difest_iso.hip: inside blomgpu_step k_cmn_bfsqf runs beside them)
enum { CM_DELP, CM_BFSQ, CM_SLS2, CM_GAM_ALE, CM_GAM = 9, CM_NSLOT };   // (the hybrid coordinate's kernel keeps its slot)

#define COLUMN_IJ(V)                                                     \
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;                  \
  if (t_ >= (V).nplane) return;                                          \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);  \
  const size_t c = t_

// ---- :61-227 ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cmn_bfsqf(const DevView *__restrict__ Vp, int n, int nn) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t p = V.f[F_p] + c, temp = V.f[F_temp] + c + (size_t)nn * np, saln = V.f[F_saln] + c + (size_t)nn * np;
  gcd_t dp = V.f[F_dp] + c + (size_t)nn * np;
  gd_t bfsqi = V.f[F_bfsqi] + c, bfsql = V.f[F_bfsql] + c, bfsqf = V.f[F_bfsqf] + c;
  gd_t gam = WK(V, CM_GAM) + c;
#define L(a, k) (a)[(size_t)((k)-1) * np]          /* Fortran level k */
  const double b1 = .5 * GRAV * GRAV * (eos::rho(L(p, 2), L(temp, 2), L(saln, 2)) - eos::rho(L(p, 2), L(temp, 1), L(saln, 1))) /
                    (L(dp, 1) + L(dp, 2));
  L(bfsqi, 1) = b1; L(bfsqi, 2) = b1; L(bfsql, 1) = b1; L(bfsql, 2) = b1;
  const int kfpl = V.m[I_kfpla][c + (size_t)(n - 1) * np];
  if (kfpl > kk) {                                                         // mixed layer down to the bottom, :105-115
    for (int k = 3; k <= kk; k++) { L(bfsqi, k) = b1; L(bfsql, k) = b1; }
    L(bfsqi, kk + 1) = b1;
    for (int k = 1; k <= kk + 1; k++) L(bfsqf, k) = BFSQMN;
    return;
  }
  const double pbot = L(p, kk + 1);
  const double pml = fmax2(.5 * (L(p, 3) + L(p, 1)), .5 * (3. * L(p, 3) - L(p, kfpl + 1)));
  const double dml = pml - L(p, 1);
  L(bfsqi, kfpl - 1) = L(bfsqi, 2);
  // One pass down the column does :117-199: the layer quantities delp, bfsq, sls2 of a level (the reference's work arrays)
  // stay in registers for the two levels that still need them; the interface value of level k-1 is averaged into
  // bfsql(k-1) and the forward elimination of the filter's tridiagonal system is done for level k-1 as soon as level
  // k's delp is known.  Only gam goes through memory (to the back substitution).  COLUMN_U levels' loads in flight.
  double delp_mm = 0., delp_m = dml, sls2_mm = 0., sls2_m, bfsq_m = BFSQMN, bi_m = b1;
  {
    const double q = fmax2(SLS0, dml * SLSMFQ);
    sls2_m = q * q;
  }
  double pup = pml, tup = L(temp, 2), sup = L(saln, 2);
  double pk = L(p, kfpl);
  double ctd = 0., btd, bei = 0., fprev = 0., bi_kfpl = 0., bl_kfpl = 0.;
  for (int k0 = kfpl; k0 <= kk; k0 += COLUMN_U) {
    double a0[COLUMN_U], a1[COLUMN_U], a2[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int kq = k0 + u <= kk ? k0 + u : kk;
      a0[u] = L(p, kq + 1); a1[u] = L(temp, kq); a2[u] = L(saln, kq);
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 + u;
      if (k > kk) break;
      const double pk1 = a0[u];
      double delp_k, bfsq_k, sls2_k, bi;
      if (pbot - pk < EPSILP) {
        delp_k = ONEMM;
        bi = bi_m;
        bfsq_k = BFSQMN;
        double q = exp_libm(-(pbot - pml) / (SLSELS * dml));
        q = fmax2(SLS0, dml * SLSMFQ * q + SLS0 * (1. - q));
        sls2_k = q * q;
      } else {
        const double plo = pbot - pk1 < EPSILP ? pbot : .5 * (pk + pk1);
        const double tlo = a1[u], slo = a2[u];
        const double dk = fmax2(ONEMM, plo - pup);
        delp_k = dk;
        bi = GRAV * GRAV * (eos::rho(pk, tlo, slo) - eos::rho(pk, tup, sup)) / dk;
        bfsq_k = fmax2(BFSQMN, bi);
        bi = bi * dk / fmax2(ONEM, dk);
        if (pbot - pk < ONEM) bi = bi_m;
        double q = exp_libm(-(pk - pml) / (SLSELS * dml));
        q = fmax2(SLS0, dml * SLSMFQ * q + SLS0 * (1. - q));
        sls2_k = q * q;
        pup = plo; tup = tlo; sup = slo;
      }
      L(bfsqi, k) = bi;
      if (k == kfpl) {
        bi_kfpl = bi;
        // first row of the tridiagonal system, level kfpl-1, :170-183
        ctd = -2. * sls2_m / (delp_m * (delp_m + delp_k));
        btd = 1. - ctd;
        bei = 1. / btd;
        fprev = bfsq_m * bei;
        L(bfsqf, k - 1) = fprev;
      } else {
        const double bl = .5 * (bi_m + bi);                                                       // :157-159
        L(bfsql, k - 1) = bl;
        if (k == kfpl + 1) bl_kfpl = bl;
        // forward elimination of level k-1, :184-196
        const double atd = -2. * sls2_mm / (delp_m * (delp_mm + delp_m));
        const double g = ctd * bei;                  // ctd of level k-2
        L(gam, k - 1) = g;
        ctd = -2. * sls2_m / (delp_m * (delp_m + delp_k));
        btd = 1. - atd - ctd;
        bei = 1. / (btd - atd * g);
        fprev = (bfsq_m - atd * fprev) * bei;
        L(bfsqf, k - 1) = fprev;
      }
      delp_mm = delp_m; delp_m = delp_k; sls2_mm = sls2_m; sls2_m = sls2_k; bfsq_m = bfsq_k; bi_m = bi;
      pk = pk1;
    }
  }
  L(bfsql, kk) = bi_m;                                                                            // :160
  if (kfpl == kk) bl_kfpl = bi_m;
  for (int k = 3; k <= kfpl - 1; k++) { L(bfsqi, k) = bi_kfpl; L(bfsql, k) = bl_kfpl; }            // :161-165
  {                                                                                               // level kk, :184-196
    const double atd = -2. * sls2_mm / (delp_m * (delp_mm + delp_m));
    const double g = ctd * bei;
    L(gam, kk) = g;
    btd = 1. - atd;
    bei = 1. / (btd - atd * g);
    fprev = (bfsq_m - atd * fprev) * bei;
    L(bfsqf, kk) = fprev;
  }
  const double f_kk = fprev;
  for (int k0 = kk - 1; k0 >= kfpl - 1; k0 -= COLUMN_U) {                                          // :197-199
    double a0[COLUMN_U], a1[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int kq = k0 - u >= kfpl - 1 ? k0 - u : kfpl - 1;
      a0[u] = L(bfsqf, kq); a1[u] = L(gam, kq + 1);
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 - u;
      if (k < kfpl - 1) break;
      fprev = a0[u] - a1[u] * fprev;
      L(bfsqf, k) = fprev;
    }
  }
  for (int k = 1; k <= kfpl - 2; k++) L(bfsqf, k) = fprev;                 // = bfsqf(kfpl-1)
  L(bfsqi, kk + 1) = bi_m;
  L(bfsqf, kk + 1) = f_kk;
#undef L
}

// ---- geopotential at layer interfaces, :437-455 ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cmn_phi(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t p = V.f[F_p] + c, temp = V.f[F_temp] + c + (size_t)nn * np, saln = V.f[F_saln] + c + (size_t)nn * np;
  gcd_t dp = V.f[F_dp] + c + (size_t)nn * np;
  gd_t phi = V.f[F_phi] + c;
  double ph = phi[(size_t)kk * np], plo = p[(size_t)kk * np];
  for (int k0 = kk - 1; k0 >= 0; k0 -= COLUMN_U) {             // COLUMN_U levels' loads in flight (blomgpu_internal.h)
    double a[COLUMN_U], b[COLUMN_U], d[COLUMN_U], e[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const size_t o = (size_t)(k0 - u >= 0 ? k0 - u : 0) * np;
      a[u] = p[o]; b[u] = dp[o]; d[u] = temp[o]; e[u] = saln[o];
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 - u;
      if (k >= 0) {
        if (!(b[u] < EPSILP)) ph = ph - eos::p_alpha(plo, a[u], d[u], e[u]);
        phi[(size_t)k * np] = ph;
        plo = a[u];
      }
    }
  }
}

// ---- slope of the local neutral surface at u- (blockIdx.y = 0) and v-points (1), :465-641 -----------------------------
// NB interfaces' loads in flight in the interior sweep.  With 4 the kernel needs 164 VGPRs: three wavefronts a SIMD, 3 072 of the
// channel's 3 442 u- and v-column wavefronts start at once and the launch lasts two rounds; NB = 2 keeps it under 128 (round 6).
template <int NB>
__global__ __launch_bounds__(64) void k_cmn_nslope(const DevView *__restrict__ Vp, int n, int nn) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  const bool isv = blockIdx.y == 1;
  if (isv ? (j < 0 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_iv][c])
          : (j < -1 || j > V.jj + 2 || i < 0 || i > V.ii + 2 || !V.m[I_iu][c])) return;
  const size_t np = V.nplane, a_ = isv ? c - V.ni : c - 1, b_ = c;      // scalar points (i-1,j)|(i,j-1) and (i,j)
  const int kk = V.kk;
  gcd_t p = V.f[F_p], phi = V.f[F_phi], bf = V.f[F_bfsqf];
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np, dp = V.f[F_dp] + (size_t)nn * np;
  gd_t nslp = (isv ? V.f[F_nslpy] : V.f[F_nslpx]) + c, nnslp = (isv ? V.f[F_nnslpy] : V.f[F_nnslpx]) + c;
  const double sci = (isv ? V.f[F_scvyi] : V.f[F_scuxi])[c];
#define A(f, x, k) (f)[(x) + (size_t)((k)-1) * np]
#define O(f, k) (f)[(size_t)((k)-1) * np]
  for (int k = 1; k <= kk; k++) { O(nslp, k) = 0.; O(nnslp, k) = 0.; }
  const int kfa = V.m[I_kfpla][a_ + (size_t)(n - 1) * np], kfb = V.m[I_kfpla][b_ + (size_t)(n - 1) * np];
  if (!(kfa <= kk || kfb <= kk)) return;
  int kmax = 1;
  for (int k0 = 3; k0 <= kk; k0 += COLUMN_U) {                           // (COLUMN_U levels' loads in flight)
    double da[COLUMN_U], db[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) { const int kq = k0 + u <= kk ? k0 + u : kk; da[u] = A(dp, a_, kq); db[u] = A(dp, b_, kq); }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++)
      if (k0 + u <= kk && (da[u] > EPSILP || db[u] > EPSILP)) kmax = k0 + u;
  }
  const int kintr = kfa > kfb ? kfa : kfb;
  int knnsl = 2;
  const double phba = A(phi, a_, kk + 1), phbb = A(phi, b_, kk + 1);
  {                                                                      // mixed layer base
    const double pm = .5 * (A(p, a_, 3) + A(p, b_, 3));
    const double rx = eos::rho(pm, A(temp, b_, 2), A(saln, b_, 2)) - eos::rho(pm, A(temp, a_, 2), A(saln, a_, 2));
    const double px = A(phi, b_, 3) - A(phi, a_, 3);
    const double bm = .5 * (A(bf, a_, 3) + A(bf, b_, 3));
    const double s = (GRAV * rx / (RHO0 * bm) + px / GRAV) * sci;
    O(nslp, 3) = s;
    if (A(phi, b_, 3) > phba && A(phi, a_, 3) > phbb) { O(nnslp, 3) = sqrt(bm) * s; knnsl = 3; }
  }
  {                                                                      // interior interfaces
    // 4 interfaces' loads in flight; T, S of the layer above an interface are the previous interface's layer below
    const int kf = kintr + 1;
    double tbm = kf <= kmax ? A(temp, b_, kf - 1) : 0., sbm = kf <= kmax ? A(saln, b_, kf - 1) : 0.;
    double tam = kf <= kmax ? A(temp, a_, kf - 1) : 0., sam = kf <= kmax ? A(saln, a_, kf - 1) : 0.;
    for (int k0 = kf; k0 <= kmax; k0 += NB) {
      double pa[NB], pb[NB], tb[NB], sb[NB], ta[NB], sa[NB], fb[NB], fa[NB], ba[NB], bb[NB];
#pragma unroll
      for (int u = 0; u < NB; u++) {
        const int kq = k0 + u <= kmax ? k0 + u : kmax;
        pa[u] = A(p, a_, kq); pb[u] = A(p, b_, kq);
        tb[u] = A(temp, b_, kq); sb[u] = A(saln, b_, kq); ta[u] = A(temp, a_, kq); sa[u] = A(saln, a_, kq);
        fb[u] = A(phi, b_, kq); fa[u] = A(phi, a_, kq); ba[u] = A(bf, a_, kq); bb[u] = A(bf, b_, kq);
      }
#pragma unroll
      for (int u = 0; u < NB; u++) {
        const int k = k0 + u;
        if (k > kmax) break;
        const double pm = .5 * (pa[u] + pb[u]);
        const double rx = .5 * (eos::rho(pm, tbm, sbm) - eos::rho(pm, tam, sam) + eos::rho(pm, tb[u], sb[u]) - eos::rho(pm, ta[u], sa[u]));
        const double px = fb[u] - fa[u];
        const double bm = .5 * (ba[u] + bb[u]);
        const double s = (GRAV * rx / (RHO0 * bm) + px / GRAV) * sci;
        O(nslp, k) = s;
        if (fb[u] > phba && fa[u] > phbb) { O(nnslp, k) = sqrt(bm) * s; knnsl = k; }
        tbm = tb[u]; sbm = sb[u]; tam = ta[u]; sam = sa[u];
      }
    }
  }
  {
    const double last = O(nnslp, knnsl);
    for (int k = knnsl + 1; k <= kmax; k++) O(nnslp, k) = last;
  }
  if (kintr < kmax) {
    const double s = O(nslp, kintr + 1), ns = O(nnslp, kintr + 1);
    for (int k = 4; k <= kintr; k++) { O(nslp, k) = s; O(nnslp, k) = ns; }
  } else {
    const double s = O(nslp, 3), ns = O(nnslp, 3);
    for (int k = 4; k <= kmax; k++) { O(nslp, k) = s; O(nnslp, k) = ns; }
  }
#undef A
#undef O
}

// ---- cmnfld1 for isopyc_bulkml = cmnfld_z: depth of the layer interfaces, thickness of the layers, :885-921 ----------------
// (the mixed-layer-depth estimates of :1110-1150 belong to the other vertical coordinates and to diagnostics)
__global__ __launch_bounds__(64) void k_cmn_z(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t p = V.f[F_p] + c, temp = V.f[F_temp] + c + (size_t)mm * np, saln = V.f[F_saln] + c + (size_t)mm * np;
  gcd_t dp = V.f[F_dp] + c + (size_t)mm * np;
  gd_t z = V.f[F_z] + c, dz = V.f[F_dz] + c;
  double zlo = -V.f[F_phi][c + (size_t)kk * np] / GRAV, plo = p[(size_t)kk * np];
  z[(size_t)kk * np] = zlo;
  for (int k0 = kk - 1; k0 >= 0; k0 -= COLUMN_U) {             // COLUMN_U levels' loads in flight (blomgpu_internal.h)
    double a[COLUMN_U], b[COLUMN_U], d[COLUMN_U], e[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const size_t o = (size_t)(k0 - u >= 0 ? k0 - u : 0) * np;
      a[u] = p[o]; b[u] = dp[o]; d[u] = temp[o]; e[u] = saln[o];
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 - u;
      if (k < 0) break;
      const double zk = b[u] < EPSILP ? zlo : zlo + eos::p_alpha(plo, a[u], d[u], e[u]) / GRAV;
      z[(size_t)k * np] = zk;
      dz[(size_t)k * np] = zlo - zk;
      zlo = zk; plo = a[u];
    }
  }
}

// ---- the branches of the vertical coordinates other than isopyc_bulkml (round 3) ---------------------------------------------
// cmnfld_bfsqf_ale, :229-350: as the isopycnic form without the mixed-layer treatment; delp and bfsq of a level go through two
// work planes to the tridiagonal filter.  (bfsqi and bfsql are zeroed everywhere first, :247-248: the launcher does that.)
// One pass down the column does what the reference does in four loops (:254-333): the interface values of level k, the layer
// mean of level k-1 and the forward elimination of row k-1 of the filter's tridiagonal system -- which needs delp(k-2..k) and
// bfsq(k-1), all still in registers (delp and bfsq are no work planes any more) -- then the back substitution.  Statements as in
// the reference; the next four levels' p, T, S are loaded ahead.
__global__ __launch_bounds__(64) void k_cmn_bfsqf_ale(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t __restrict__ p = V.f[F_p] + c, temp = V.f[F_temp] + c + (size_t)nn * np, saln = V.f[F_saln] + c + (size_t)nn * np;
  gd_t __restrict__ bfsqi = V.f[F_bfsqi] + c, bfsql = V.f[F_bfsql] + c, bfsqf = V.f[F_bfsqf] + c;
  gd_t __restrict__ gam = WK(V, CM_GAM_ALE) + c;
#define L(a, k) (a)[(size_t)((k)-1) * np]
  const double sls2 = SLS0 * SLS0, pbot = L(p, kk + 1);
  double pk = L(p, 2);                                             // p(k) of the level being worked on
  double pup = .5 * (L(p, 1) + pk), tup = L(temp, 1), sup = L(saln, 1);
  double bi_prev = BFSQMN;                                         // bfsqi(k-1); bfsqi(1) = bfsqmn while the loop runs (:256)
  double d_m2 = 0., d_m1 = V.f[F_dp][c + (size_t)nn * np], d_0 = 0.;   // delp(k-2), delp(k-1), delp(k); delp(1) = dp(1) (:300)
  double q_m1 = 0.;                                                // bfsq(k-1)
  double bei = 0., f_prev = 0., ctd_prev = 0.;                     // of the last eliminated row
  for (int k0 = 2; k0 <= kk; k0 += 4) {
    double a_p[4], a_t[4], a_s[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int kq = k0 + u <= kk ? k0 + u : kk;
      a_p[u] = L(p, kq + 1); a_t[u] = L(temp, kq); a_s[u] = L(saln, kq);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u;
      if (k > kk) break;
      const double pk1 = a_p[u];
      double bi, bq;
      if (pbot - pk < EPSILP) {                                    // :259-263
        d_0 = ONEMM;
        bi = bi_prev;
        bq = BFSQMN;
      } else {                                                     // :264-289
        const double plo = pbot - pk1 < EPSILP ? pbot : .5 * (pk + pk1);
        const double tlo = a_t[u], slo = a_s[u];
        const double dk = fmax2(ONEMM, plo - pup);
        d_0 = dk;
        bi = GRAV * GRAV * (eos::rho(pk, tlo, slo) - eos::rho(pk, tup, sup)) / dk;
        bq = fmax2(BFSQMN, bi);
        bi = bi * dk / fmax2(ONEM, dk);
        if (pbot - pk < ONEM) bi = bi_prev;
        pup = plo; tup = tlo; sup = slo;
      }
      L(bfsqi, k) = bi;
      if (k == 2) {
        // bfsqi(1) = bfsqi(2), bfsq(1) from it (:301-302); bfsql(1); row 1 of the system (:321-322)
        L(bfsqi, 1) = bi;
        q_m1 = fmax2(BFSQMN, bi);
        L(bfsql, 1) = .5 * (bi + bi);
        ctd_prev = -2. * sls2 / (d_m1 * (d_m1 + d_0));
        bei = 1. / (1. - ctd_prev);
        f_prev = q_m1 * bei;
        L(bfsqf, 1) = f_prev;
      } else {
        L(bfsql, k - 1) = .5 * (bi_prev + bi);                     // :304-306
        // row k-1 (:323-332): delp(k-2), delp(k-1), delp(k) = d_m2, d_m1, d_0
        const double g = ctd_prev * bei;
        L(gam, k - 1) = g;
        const double at = -2. * sls2 / (d_m1 * (d_m2 + d_m1));
        const double ct = -2. * sls2 / (d_m1 * (d_m1 + d_0));
        bei = 1. / ((1. - at - ct) - at * g);
        f_prev = (q_m1 - at * f_prev) * bei;
        L(bfsqf, k - 1) = f_prev;
        ctd_prev = ct;
      }
      d_m2 = d_m1; d_m1 = d_0; q_m1 = bq; bi_prev = bi; pk = pk1;
    }
  }
  L(bfsql, kk) = bi_prev;                                          // :307
  {                                                                // row kk
    const double g = ctd_prev * bei;
    L(gam, kk) = g;
    const double at = -2. * sls2 / (d_m1 * (d_m2 + d_m1));
    bei = 1. / ((1. - at) - at * g);
    f_prev = (q_m1 - at * f_prev) * bei;
    L(bfsqf, kk) = f_prev;
  }
  double fn = f_prev;
  for (int k0 = kk - 1; k0 >= 1; k0 -= 4) {                        // back substitution, :333
    double a_f[4], a_g[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int kq = k0 - u >= 1 ? k0 - u : 1; a_f[u] = L(bfsqf, kq); a_g[u] = L(gam, kq + 1); }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 - u;
      if (k < 1) break;
      fn = a_f[u] - a_g[u] * fn;
      L(bfsqf, k) = fn;
    }
  }
  L(bfsqi, kk + 1) = bi_prev;
  L(bfsqf, kk + 1) = f_prev;
#undef L
}

// cmnfld_bfsqi_ale, :352-421: p from dp (j,i = -2..+3), then the interface buoyancy frequency (j,i = 0..+1)
__global__ __launch_bounds__(64) void k_cmn_bfsqi_ale(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t p = V.f[F_p] + c, temp = V.f[F_temp] + c + (size_t)nn * np, saln = V.f[F_saln] + c + (size_t)nn * np;
  gd_t bfsqi = V.f[F_bfsqi] + c;
#define L(a, k) (a)[(size_t)((k)-1) * np]
  const double pbot = L(p, kk + 1);
  // the level above travels in registers (bfsqi(1) = bfsqmn while the loop runs), the next four levels' p, T, S are loaded ahead
  double pk = L(p, 2);
  double pup = .5 * (L(p, 1) + pk), tup = L(temp, 1), sup = L(saln, 1);
  double bi_prev = BFSQMN, bi2 = BFSQMN;
  for (int k0 = 2; k0 <= kk; k0 += 4) {
    double a_p[4], a_t[4], a_s[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int kq = k0 + u <= kk ? k0 + u : kk;
      a_p[u] = L(p, kq + 1); a_t[u] = L(temp, kq); a_s[u] = L(saln, kq);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u;
      if (k > kk) break;
      const double pk1 = a_p[u];
      double bi;
      if (pbot - pk < EPSILP) bi = bi_prev;
      else {
        const double plo = pbot - pk1 < EPSILP ? pbot : .5 * (pk + pk1);
        const double tlo = a_t[u], slo = a_s[u];
        bi = GRAV * GRAV * (eos::rho(pk, tlo, slo) - eos::rho(pk, tup, sup)) / fmax2(ONEM, plo - pup);
        if (pbot - pk < ONEM) bi = bi_prev;
        pup = plo; tup = tlo; sup = slo;
      }
      L(bfsqi, k) = bi;
      if (k == 2) bi2 = bi;
      bi_prev = bi; pk = pk1;
    }
  }
  L(bfsqi, 1) = bi2;
  L(bfsqi, kk + 1) = bi_prev;
#undef L
}

// cmnfld_nnslope_ale, :813-883: with ltedtp = 'neutral' the slopes nslpx, nslpy are those the neutral diffusion found between
// the columns (stage_ndiff.hip); here they are multiplied with the buoyancy frequency as far down as both columns reach
__global__ __launch_bounds__(64) void k_cmn_nnslope_ale(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  const bool isv = blockIdx.y == 1;
  if (isv ? (j < 0 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_iv][c])
          : (j < -1 || j > V.jj + 2 || i < 0 || i > V.ii + 2 || !V.m[I_iu][c])) return;
  const size_t np = V.nplane, a_ = isv ? c - V.ni : c - 1, b_ = c;
  const int kk = V.kk;
  gcd_t p = V.f[F_p], bf = V.f[F_bfsqf];
  gcd_t nslp = (isv ? V.f[F_nslpy] : V.f[F_nslpx]) + c;
  gd_t nnslp = (isv ? V.f[F_nnslpy] : V.f[F_nnslpx]) + c;
  const double pba = p[a_ + (size_t)kk * np], pbb = p[b_ + (size_t)kk * np];
  int knnsl = 1;
  double last = 0.;
  nnslp[0] = 0.;
  for (int k = 2; k <= kk; k++) {
    if (p[b_ + (size_t)(k - 1) * np] < pba && p[a_ + (size_t)(k - 1) * np] < pbb) {
      const double bfsqm = .5 * (bf[a_ + (size_t)(k - 1) * np] + bf[b_ + (size_t)(k - 1) * np]);
      last = sqrt(bfsqm) * nslp[(size_t)(k - 1) * np];
      nnslp[(size_t)(k - 1) * np] = last;
      knnsl = k;
    } else break;
  }
  for (int k = knnsl + 1; k <= kk; k++) nnslp[(size_t)(k - 1) * np] = last;
}

// cmnfld_nslope_ale, :654-811 (after the geopotential, which is k_cmn_phi's loop): u- (blockIdx.y = 0) and v-points (1)
__global__ __launch_bounds__(64) void k_cmn_nslope_ale(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  const bool isv = blockIdx.y == 1;
  if (isv ? (j < 0 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_iv][c])
          : (j < -1 || j > V.jj + 2 || i < 0 || i > V.ii + 2 || !V.m[I_iu][c])) return;
  const size_t np = V.nplane, a_ = isv ? c - V.ni : c - 1, b_ = c;
  const int kk = V.kk;
  gcd_t p = V.f[F_p], phi = V.f[F_phi], bf = V.f[F_bfsqf];
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np, dp = V.f[F_dp] + (size_t)nn * np;
  gd_t nslp = (isv ? V.f[F_nslpy] : V.f[F_nslpx]) + c, nnslp = (isv ? V.f[F_nnslpy] : V.f[F_nnslpx]) + c;
  const double sci = (isv ? V.f[F_scvyi] : V.f[F_scuxi])[c];
#define A(f, x, k) (f)[(x) + (size_t)((k)-1) * np]
#define O(f, k) (f)[(size_t)((k)-1) * np]
  for (int k = 1; k <= kk; k++) { O(nslp, k) = 0.; O(nnslp, k) = 0.; }
  int kmax = 1;
  for (int k0 = 2; k0 <= kk; k0 += 8) {                                  // eight levels' loads in flight
    double d0[8], d1[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int kq = k0 + u <= kk ? k0 + u : kk; d0[u] = A(dp, a_, kq); d1[u] = A(dp, b_, kq); }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (k0 + u <= kk && (d0[u] > EPSILP || d1[u] > EPSILP)) kmax = k0 + u;
  }
  int knnsl = 2;
  const double phba = A(phi, a_, kk + 1), phbb = A(phi, b_, kk + 1);
  {
    // 4 interfaces' loads in flight; T, S of the layer above an interface are the previous interface's layer below
    double tbm = A(temp, b_, 1), sbm = A(saln, b_, 1), tam = A(temp, a_, 1), sam = A(saln, a_, 1);
    for (int k0 = 2; k0 <= kmax; k0 += 4) {
      double pa[4], pb[4], tb[4], sb[4], ta[4], sa[4], fb[4], fa[4], ba[4], bb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int kq = k0 + u <= kmax ? k0 + u : kmax;
        pa[u] = A(p, a_, kq); pb[u] = A(p, b_, kq);
        tb[u] = A(temp, b_, kq); sb[u] = A(saln, b_, kq); ta[u] = A(temp, a_, kq); sa[u] = A(saln, a_, kq);
        fb[u] = A(phi, b_, kq); fa[u] = A(phi, a_, kq); ba[u] = A(bf, a_, kq); bb[u] = A(bf, b_, kq);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u;
        if (k > kmax) break;
        const double pm = .5 * (pa[u] + pb[u]);
        const double rx = .5 * (eos::rho(pm, tbm, sbm) - eos::rho(pm, tam, sam) + eos::rho(pm, tb[u], sb[u]) - eos::rho(pm, ta[u], sa[u]));
        const double px = fb[u] - fa[u];
        const double bm = .5 * (ba[u] + bb[u]);
        const double s = (GRAV * rx / (RHO0 * bm) + px / GRAV) * sci;
        O(nslp, k) = s;
        if (fb[u] > phba && fa[u] > phbb) { O(nnslp, k) = sqrt(bm) * s; knnsl = k; }
        tbm = tb[u]; sbm = sb[u]; tam = ta[u]; sam = sa[u];
      }
    }
  }
  for (int k = knnsl + 1; k <= kmax; k++) O(nnslp, k) = O(nnslp, knnsl);
#undef A
#undef O
}

// p(k+1) = p(k) + dp(kn), j,i = -2..+3 (cmnfld_bfsqi_ale :364-375)
__global__ __launch_bounds__(64) void k_cmn_pscan3(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  if (j < -2 || j > V.jj + 3 || i < -2 || i > V.ii + 3 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  double a = V.f[F_p][c];
  for (int k = 0; k < V.kk; k++) { a = a + V.f[F_dp][c + (size_t)(k + nn) * np]; V.f[F_p][c + (size_t)(k + 1) * np] = a; }
}

int st_cmnfld_bfsqi_ale(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "cmnfld_bfsqi_ale: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:196-212)");
  const dim3 g1 = plane_grid(h, 1, 64);
  hipLaunchKernelGGL(k_cmn_pscan3, g1, dim3(64), 0, c->stream, c->d, nn);
  HIPCHK(c, hipMemsetAsync(h.f[F_bfsqi], 0, sizeof(double) * (size_t)(h.kk + 1) * h.nplane, c->stream));        // :378
  hipLaunchKernelGGL(k_cmn_bfsqi_ale, g1, dim3(64), 0, c->stream, c->d, nn);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// cmnfld_mldl82, :933-995 (mixed layer depth after Levitus 1982) and the selection mld = mldl82, dpml = dpmll82 of cmnfld1 for
// the vertical coordinates other than isopyc_bulkml with mldmth = 'lev82' (:1121-1131)
#define DBCL82 .0003
#define ONECM 98.06
__global__ __launch_bounds__(64) void k_cmn_mldl82(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  COLUMN_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  gcd_t p = V.f[F_p] + c, z = V.f[F_z] + c, dz = V.f[F_dz] + c;
  gcd_t temp = V.f[F_temp] + c + (size_t)mm * np, saln = V.f[F_saln] + c + (size_t)mm * np, dp = V.f[F_dp] + c + (size_t)mm * np;
#define L(a, k) (a)[(size_t)((k)-1) * np]
  int k = 2;
  double pup = L(p, 1) + .5 * L(dp, 1), zup = L(z, 1) + .5 * L(dz, 1), dbup = 0., mld, dpml;
  const double t1 = L(temp, 1), s1 = L(saln, 1);
  while (true) {
    if (L(dp, k) > ONECM) {
      const double plo = L(p, k) + .5 * L(dp, k), zlo = L(z, k) + .5 * L(dz, k);
      const double dblo = GRAV * (1. - eos::rho(plo, t1, s1) / eos::rho(plo, L(temp, k), L(saln, k)));
      if (dblo <= DBCL82) { pup = plo; zup = zlo; dbup = dblo; }
      else {
        dbup = fmin2(dbup, DBCL82 - EPSILP);
        mld = (zup * (dblo - DBCL82) + zlo * (DBCL82 - dbup)) / (dblo - dbup) - L(z, 1);
        dpml = (pup * (dblo - DBCL82) + plo * (DBCL82 - dbup)) / (dblo - dbup) - L(p, 1);
        break;
      }
    }
    k = k + 1;
    if (k > kk) {
      mld = L(z, kk + 1) - L(z, 1);
      dpml = L(p, kk + 1) - L(p, 1);
      break;
    }
  }
#undef L
  V.f[F_mldl82][c] = mld;
  V.f[F_mld][c] = mld;
  V.f[F_dpml][c] = dpml;
}

int st_cmnfld1(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)nn; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  TimeScope ts(c, "cmnfld");
  hipLaunchKernelGGL(k_cmn_z, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, mm);
  // the other vertical coordinates also need the mixed layer depth (mldmth = 'lev82', the default; 'boy04' is not built)
  if (h.P.vcoord_tag != 1) hipLaunchKernelGGL(k_cmn_mldl82, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, mm);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// cmnfld2(m,n,mm,nn,k1m,k1n) for isopyc_bulkml; the slope part only with eitmth = 'gm' (:1208-1235; edritp is not carried)
int st_cmnfld2(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.nwk < CM_NSLOT) return ctx_fail(c, "cmnfld2: device work space too small");
  {
    double *ptrs[2] = {h.f[F_temp], h.f[F_saln]};                                           // :1171-1172
    const int nl[2] = {2 * h.kk, 2 * h.kk}, it[2] = {1, 1};
    if (int rc = st_xctilr_multi(c, 2, ptrs, nl, 3, 3, it)) return rc;
  }
  const dim3 g1 = plane_grid(h, 1, 64), g2 = plane_grid(h, 2, 64);
  if (h.P.vcoord_tag != 1) {
    // the other vertical coordinates, :1203-1234: the filtered buoyancy frequency always, the slopes with eitmth = 'gm'
    if (h.P.ltedtp_opt == 2 && h.P.eitmth == 2) {                                           // :817-818
      if (int rc = st_xctilr(c, h.f[F_nslpx], 1, h.kk, 2, 2, 13)) return rc;
      if (int rc = st_xctilr(c, h.f[F_nslpy], 1, h.kk, 2, 2, 14)) return rc;
    }
    TimeScope ts(c, "cmnfld");
    HIPCHK(c, hipMemsetAsync(h.f[F_bfsqi], 0, sizeof(double) * (size_t)(h.kk + 1) * h.nplane, c->stream));      // :247-248
    HIPCHK(c, hipMemsetAsync(h.f[F_bfsql], 0, sizeof(double) * (size_t)h.kk * h.nplane, c->stream));
    hipLaunchKernelGGL(k_cmn_bfsqf_ale, g1, dim3(64), 0, c->stream, c->d, nn);
    if (h.P.eitmth == 2) {
      if (h.P.ltedtp_opt == 2) hipLaunchKernelGGL(k_cmn_nnslope_ale, g2, dim3(64), 0, c->stream, c->d);   // :1229-1233
      else {
        hipLaunchKernelGGL(k_cmn_phi, g1, dim3(64), 0, c->stream, c->d, nn);
        hipLaunchKernelGGL(k_cmn_nslope_ale, g2, dim3(64), 0, c->stream, c->d, nn);
      }
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  if (int rc = st_kfpla_halo(c, n)) return rc;                                              // :1176-1196
  if (h.P.eitmth != 2) return 0;
  // Inside blomgpu_step (phys_dag): the three kernels on the second stream.  They read T, S, dp, p, kfpla and write bfsqi, bfsql, bfsqf, phi,
  // nslpx/y, nnslpx/y and work slot CM_GAM; what follows on the context's stream until something waits for them -- difest_isobml's halo
  // updates (u, v, ubflxs_p, pbu ..), its common part and its vertical chain (stage_difest_iso.hip) -- touches none of these.  The first
  // reader of the slopes is difest's lateral part, which st_difest_isobml puts behind them on the same stream; blomgpu_step joins
  // before eddtra otherwise.
  const bool aside = ctx_overlap_on(c) && (c->phys_dag & 1) && !c->tiling.multi();
  hipStream_t s = c->stream;
  if (aside) {
    // p is read by the three kernels and rewritten -- with the same values when the state is consistent, out to ii+3 -- by the pressure scan
    // at the start of difest_isobml (api.hip: blomgpu_halo_difest; nothing in between changes dp or its halo): that scan goes in front of the
    // fork, so that no kernel writes p while these read it
    if (int rc = launch_pscan(c, nn, -2, 3)) return rc;
    c->pscan_done_ahead = true;
    if (int rc = ctx_side_fork(c, 4)) return rc;
    s = c->side;
  }
  TimeScope ts(c, "cmnfld");
  hipLaunchKernelGGL(k_cmn_bfsqf, g1, dim3(64), 0, s, c->d, n, nn);
  hipLaunchKernelGGL(k_cmn_phi, g1, dim3(64), 0, s, c->d, nn);
  switch (c->cmn_nslope_nb) {
    case 4: hipLaunchKernelGGL(k_cmn_nslope<4>, g2, dim3(64), 0, s, c->d, n, nn); break;
    case 3: hipLaunchKernelGGL(k_cmn_nslope<3>, g2, dim3(64), 0, s, c->d, n, nn); break;
    default: hipLaunchKernelGGL(k_cmn_nslope<2>, g2, dim3(64), 0, s, c->d, n, nn); break;
  }
  HIPCHK(c, hipGetLastError());
  c->cmn_on_side = aside;
  return 0;
}
