// mxlayr -- the bulk mixed layer of the isopycnic coordinate: turbulent kinetic energy balance of the mixed layer
// (Oberhuber 1993 closure + restratification by mixed layer eddies, Fox-Kemper et al. 2008), detrainment into /
// entrainment from the isopycnic layers, surface forcing (heat, salt, brine plumes, penetrating shortwave), then the new
// layer structure at the velocity points -- phy/mod_mxlayr.F90:130-1429, called after thermf for vcoord_type =
// 'isopyc_bulkml' (phy/mod_blom_step.F90:188-192).
//
// Kernels
//   k_mxl_bg2_sig, k_mxl_bg2_grad, k_mxl_bg2_sum   squared lateral buoyancy gradient of the mixed layer into util1
//                                                  (:222-278; util2, util3 hold its u- and v-point parts as in the reference)
//   k_mxl_column                                   :293-1241, one thread per column.  The reference copies a column into 1-D
//                                                  arrays, edits them and copies them back; here the 1-D arrays ARE the
//                                                  column's planes, edited in place (as in stage_convec.hip), the interface
//                                                  pressures of the column (pres) and the brine weights (bc) live in two
//                                                  (kk+1)-level work fields.  The TKE balance iterations and the forcing touch
//                                                  a few layers around the mixed layer base; what every column pays is the
//                                                  pressure scan at the start and the sweep of the copy-back rules at the end
//                                                  (negative salinity / tracer clamps with their corrections, :1219-1241).
//   tail (:1246-1374)                              k_mom_pupv (pu, pv from the old dpu, dpv), dp halo + p + dpu/dpv
//                                                  (st_mxlayr_tail), k_convec_velocity (the same conservative remap of u, v
//                                                  onto the new layers as convec's, phy/mod_convec.F90:315-391).
// exp() is the host libm's (exp_libm.h): the iterations' decisions depend on its last bit.
// Roofline: HBM; ~ (9 + 2 ntr) F for the column kernel + 14 F for the tail.
#include "blomgpu_internal.h"
#include "eos.h"
#include "diapfl_common.h"
#include "exp_libm.h"

#define GRAV 9.806
#define ALPHA0 1.e-3
#define SPCIFH 3.99e3
#define EPSILP 1.e-12
#define ONEM 9806.
#define TENCM 980.6
#define ONECM 98.06
#define ONEMM 9.806
#define ONEMU 9.806e-3

#define PLANE_IJ(V)                                                        \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_;                                                     \
  (void)i; (void)j; (void)c

// namelist-type variables of mod_mxlayr, mod_eddtra, mod_niw, mod_swabs that the column kernel reads
struct MxlPar {
  double rm0, rm5, ce, rtau, rlf, niwgf, niwbf, swamxd, mltmin, thktop;
  int rtsflg;
  int *maxitr_count;     // diagnostic: columns whose TKE-balance iteration ended at maxitr (the reference prints them and goes on)
};

namespace eos0 {
using namespace eos;
constexpr double alpha0 = 1.e-3;
constexpr double ap110 = a11 - a21 / alpha0, ap120 = a12 - a22 / alpha0, ap130 = a13 - a23 / alpha0, ap140 = a14 - a24 / alpha0,
                 ap150 = a15 - a25 / alpha0, ap160 = a16 - a26 / alpha0;
// dsigdt0 / dsigds0, phy/mod_eos.F90:263-282, :325-344 (surface reference pressure: coefficients :118-129)
__device__ inline double dsigdt0(double th, double s) {
  const double r1 = ap110 + (ap120 + ap140 * th + ap150 * s) * th + (ap130 + ap160 * s) * s;
  const double r2i = 1. / (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s);
  return (ap120 + 2. * ap140 * th + ap150 * s - (a22 + 2. * a24 * th + a25 * s) * r1 * r2i) * r2i;
}
__device__ inline double dsigds0(double th, double s) {
  const double r1 = ap110 + (ap120 + ap140 * th + ap150 * s) * th + (ap130 + ap160 * s) * s;
  const double r2i = 1. / (a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s);
  return (ap130 + ap150 * th + 2. * ap160 * s - (a23 + a25 * th + 2. * a26 * s) * r1 * r2i) * r2i;
}
// p_p_alpha(p1,p2,th,s), phy/mod_eos.F90:430-476
__device__ inline double p_p_alpha(double p1, double p2, double th, double s) {
  const double r1_3 = 1. / 3., r1_5 = 1. / 5., r1_7 = 1. / 7., r1_9 = 1. / 9., r1_10 = 1. / 10.;
  const double a1 = a11 + (a12 + a14 * th + a15 * s) * th + (a13 + a16 * s) * s;
  const double a2 = a21 + (a22 + a24 * th + a25 * s) * th + (a23 + a26 * s) * s;
  const double b1 = b11 + b12 * th + b13 * s;
  const double b2 = b21 + b22 * th + b23 * s;
  const double pm = .5 * (p2 + p1);
  const double dp = .5 * (p2 - p1);
  const double r = dp / (a1 + b1 * pm);
  const double q = b1 * r;
  return 2. * dp * r *
         (a2 + b2 * pm +
          (a2 - a1 * b2 / b1) * q *
              (r1_3 + q * (r1_3 + q * (r1_5 + q * (r1_5 + q * (r1_7 + q * (r1_7 + q * (r1_9 + q * (r1_9 + q * r1_10)))))))));
}
}  // namespace eos0

// ---- :222-235 buoyancy of the mixed layer at the surface reference pressure -------------------------------------------
__global__ void k_mxl_bg2_sig(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, o1 = c + (size_t)nn * np, o2 = o1 + np;
  const double d1 = V.f[F_dp][o1], d2 = V.f[F_dp][o2];
  const double q = 1. / (d1 + d2);
  const double tmxl = (V.f[F_temp][o1] * d1 + V.f[F_temp][o2] * d2) * q;
  const double smxl = (V.f[F_saln][o1] * d1 + V.f[F_saln][o2] * d2) * q;
  V.f[F_util1][c] = GRAV * ALPHA0 * eos::sig0(tmxl, smxl);
}

// ---- :237-256 its squared gradient at u-points (i = 1..ii+1) and v-points (j = 1..jj+1) --------------------------------
__global__ void k_mxl_bg2_grad(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  gcd_t u1 = V.f[F_util1];
  if (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1 && V.m[I_iu][c]) {
    const double q = (u1[c] - u1[c - 1]) * V.f[F_scuxi][c];
    V.f[F_util2][c] = q * q;
  }
  if (j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii && V.m[I_iv][c]) {
    const double q = (u1[c] - u1[c - V.ni]) * V.f[F_scvyi][c];
    V.f[F_util3][c] = q * q;
  }
}

// ---- :257-278 averaged onto the p-points ---------------------------------------------------------------------------------
__global__ void k_mxl_bg2_sum(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  gci_t ip = V.m[I_ip];
  gcd_t u2 = V.f[F_util2], u3 = V.f[F_util3];
  const double slbg0 = 0.;
  double r;
  if (ip[c - 1] + ip[c + 1] == 2) r = .5 * (u2[c] + u2[c + 1]);
  else if (ip[c - 1] == 1) r = u2[c];
  else if (ip[c + 1] == 1) r = u2[c + 1];
  else r = 0.;
  if (ip[c - V.ni] + ip[c + V.ni] == 2) r = r + .5 * (u3[c] + u3[c + V.ni]);
  else if (ip[c - V.ni] == 1) r = r + u3[c];
  else if (ip[c + V.ni] == 1) r = r + u3[c + V.ni];
  V.f[F_util1][c] = r + slbg0;
}

// ---- :293-1241 the column ------------------------------------------------------------------------------------------------
#define MAXITR 20
// A loop over the tracers of a column with the loads of four tracers in flight: load(nt, x, y, z) reads what tracer nt needs,
// use(nt, x, y, z) computes and stores.  (One tracer per iteration is one exposed memory latency per tracer and loop -- with 24
// tracers most of the kernel's time.)  What the loop reads besides the tracers is loaded by the caller beforehand.
template <class L, class U>
__device__ inline void tracers4(int ntr, L load, U use, int first = 0) {
  for (int nt0 = first; nt0 < ntr; nt0 += 4) {
    double x[4], y[4], z[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      x[b] = 0.; y[b] = 0.; z[b] = 0.;
      load(nt0 + b < ntr ? nt0 + b : ntr - 1, x[b], y[b], z[b]);
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (nt0 + b < ntr) use(nt0 + b, x[b], y[b], z[b]);
  }
}
#define MAXTR_MXL 64     // tracer sums of a column: dynamically indexed (private memory), any tracer count up to this
__global__ __launch_bounds__(64) void k_mxl_column(const DevView *__restrict__ Vp, MxlPar M, int n, int nn KPROF_ARGS) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int kk = V.kk, ntr = V.ntr, ni = V.ni;
  const size_t np = V.nplane;
  const Params P = V.P;
  const double delt1 = P.delt1;
  // 1-based level k of the time level n: element c + (k - 1 + nn) * np
  gd_t ttem = V.f[F_temp] + c + (size_t)nn * np - np, ssal = V.f[F_saln] + c + (size_t)nn * np - np;
  gd_t delp = V.f[F_dp] + c + (size_t)nn * np - np, dens = V.f[F_sigma] + c + (size_t)nn * np - np;
  gcd_t densr = V.f[F_sigmar] + c - np;
  gd_t trc = V.f[F_trc] + c + (size_t)nn * np - np; // tracer nt (0-based): + nt * 2 * kk * np
  gd_t pres = V.f[F_wkp1] + c - np, bc = V.f[F_wkp0] + c - np;
  gcd_t uu = V.f[F_u] + c + (size_t)nn * np - np, vv = V.f[F_v] + c + (size_t)nn * np - np;
  gcd_t dpu = V.f[F_dpu] + c + (size_t)nn * np - np, dpv = V.f[F_dpv] + c + (size_t)nn * np - np;
  const size_t ntl = (size_t)2 * kk * np;
#define TT(k) ttem[(size_t)(k) * np]
#define SS(k) ssal[(size_t)(k) * np]
#define DP(k) delp[(size_t)(k) * np]
#define DN(k) dens[(size_t)(k) * np]
#define DR(k) densr[(size_t)(k) * np]
#define PR(k) pres[(size_t)(k) * np]
#define BC(k) bc[(size_t)(k) * np]
#define TR(nt, k) trc[(size_t)(nt) * ntl + (size_t)(k) * np]
#define SIG(t, s) eos::sig(P, t, s)
  const double kappa = .4, mu = 2., ustmin = .001, mldjmp = 1.e-3;
  const double cori20 = 4.9745e-5, ci = 44. / 63.;
  const double bpdrho = .4, bpmndp = 10. * ONEM, bpmxdp = 500. * ONEM, bpdpmn = 1. * ONEM, dsgmnr = .1;
  const double mltmin = M.mltmin, thktop = M.thktop, rm5 = M.rm5;

  // KPROF words: 0 start, 1 pressure scan done, 2 detrainment's iteration done, 3 detrainment branch done (forcing, fossil layer), 4 entrainment's
  // walk done, 5 entrainment branch done, 6 end; 7 trips of the wave through the entrainment's inner iteration
  [[maybe_unused]] const int wid = blockIdx.x;
  KPROF_MARK(wid, 0);
  // 2-D inputs of the column
  const double surflx = V.f[F_surflx][c], salflx = V.f[F_salflx][c], brnflx = V.f[F_brnflx][c], sswflx = V.f[F_sswflx][c];
  const double surrlx = V.f[F_surrlx][c], salrlx = V.f[F_salrlx][c], swfc2 = V.f[F_swfc2][c], swal2 = V.f[F_swal2][c];
  const double coriop = V.f[F_coriop][c], ustar = V.f[F_ustar][c], scp2 = V.f[F_scp2][c], bg2 = V.f[F_util1][c];

  // interface pressures of the column (:296-311; the rest of the extraction is the planes themselves)
  {
    double acc = V.f[F_p][c];
    PR(1) = acc;
    for (int k0 = 1; k0 <= kk; k0 += COLUMN_U) {
      double a[COLUMN_U];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) a[u] = DP(k0 + u <= kk ? k0 + u : kk);
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++)
        if (k0 + u <= kk) { acc = acc + a[u]; PR(k0 + u + 1) = acc; }
    }
  }

  KPROF_MARK(wid, 1);
  // ---- turbulent kinetic energy balance of the mixed layer, :313-384 ------------------------------------------------------
  double q = 1. / (DP(1) + DP(2));
  double tmxl = (TT(1) * DP(1) + TT(2) * DP(2)) * q;
  double smxl = (SS(1) * DP(1) + SS(2) * DP(2)) * q;
  const double alfa = -ALPHA0 * eos0::dsigdt0(tmxl, smxl);
  const double beta = ALPHA0 * eos0::dsigds0(tmxl, smxl);
  const double bfltot = GRAV * ALPHA0 * (alfa * surflx / SPCIFH - beta * (salflx - brnflx));
  V.f[F_buoyfl][c] = bfltot;
  const double bflpsw = GRAV * ALPHA0 * alfa * swfc2 * sswflx / SPCIFH;

  double pmxl = PR(3);
  q = ALPHA0 / GRAV;
  const double lui = fabs(coriop) * q / (kappa * fmax2(ustmin, ustar));
  const double lei = 1. / (ONEM * swal2);
  const double cus = M.rm0 * V.f[F_ustar3][c];
  const double cni = M.niwgf * M.niwbf * V.f[F_idkedt][c];
  const double cbftot = .5 * bfltot * q;
  const double cbfpsw = .5 * bflpsw * q;
  double crs;
  if (M.rtsflg == 1) crs = ci * M.ce * bg2 * (q * q * q) * sqrt(scp2 / (coriop * coriop + M.rtau * M.rtau)) * M.rlf;
  else if (M.rtsflg == 2) crs = ci * M.ce * bg2 * (q * q * q) * sqrt(scp2) * M.rlf / cori20;
  else crs = ci * M.ce * bg2 * M.rlf * (q * q * q) * sqrt(scp2 / fmin2(cori20 * cori20, coriop * coriop + M.rtau * M.rtau));

  double lbi, rm1, rm2, rm3, rm4;
  double mtkeus, mtkeni, mtkebf, mtkers, mtkepe, mtkeke, tkew;
  rm1 = exp_libm(-lui * pmxl);
  q = lei * pmxl;
  rm3 = exp_libm(-q);
  rm4 = 2. / q;
  q = (cbftot - cbfpsw * (rm4 * (1. - rm3) - rm3));
  if (q < 0.) {
    lbi = lui;
    rm2 = rm1;
  } else {
    lbi = lui * kappa / mu;
    rm2 = exp_libm(-lbi * pmxl);
  }
  mtkeus = cus * rm1;
  mtkeni = cni * rm1;
  mtkebf = q * rm2 * pmxl;
  mtkers = -crs * pmxl * pmxl * pmxl;
  mtkepe = 0.;
  mtkeke = 0.;
  tkew = mtkeus + mtkeni + mtkebf + mtkers;

  const double pradd = M.swamxd * ONEM;
  int kfpl = V.m[I_kfpla][c + (size_t)(n - 1) * np];
  int k, kmax, kfmax, nitr;
  double dpmxl, tkeo = 0., dtke, tfsl, sfsl, dpfsl, dptopl, dpt, pup, plo = 0., drhup, drhlo = 0., pbrnd, bcwsum, bdpsum, tup, sup, dup, dsgdt,
         dsgds, bpc, bpmldp, pswbas, pswup, pswlo, ttmp, stmp, sigtmp, sigfsl, tmxl0, smxl0, dpe0, tdps, sdps, dpe, dps, um, vm, dke,
         dke0, tkeu, tkel = 0., uk, vk;
  double trfsl[MAXTR_MXL], trdps[MAXTR_MXL];
  // the entrainment's sums of the first four tracers in registers (trdps is indexed by a variable: private memory, a memory round trip per
  // access -- three or four of them in every trip of the entrainment loop); tracers 5.. keep to trdps
  double trd4[4] = {0., 0., 0., 0.};
  // sums += tracers of layer kq_ times w_ (every tracer: its own sum, its own order)
  auto trd_add = [&](int kq_, double w_) {
    double y_[4];
#pragma unroll
    for (int b = 0; b < 4; b++) y_[b] = TR(b < ntr ? b : ntr - 1, kq_);
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (b < ntr) trd4[b] = trd4[b] + y_[b] * w_;
    tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = trdps[nt]; y = TR(nt, kq_); },
             [&](int nt, double x, double y, double) { trdps[nt] = x + y * w_; }, 4);
  };
  // the same with the first four tracers' values of the layer already in registers
  auto trd_add_pre = [&](int kq_, double w_, const double *y_) {
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (b < ntr) trd4[b] = trd4[b] + y_[b] * w_;
    tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = trdps[nt]; y = TR(nt, kq_); },
             [&](int nt, double x, double y, double) { trdps[nt] = x + y * w_; }, 4);
  };
  // sums = tracers of layer 2 times its thickness
  auto trd_init = [&](double dp2_) {
    double x_[4];
#pragma unroll
    for (int b = 0; b < 4; b++) x_[b] = TR(b < ntr ? b : ntr - 1, 2);
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (b < ntr) trd4[b] = x_[b] * dp2_;
    tracers4(ntr, [&](int nt, double &x, double &, double &) { x = TR(nt, 2); }, [&](int nt, double x, double, double) { trdps[nt] = x * dp2_; }, 4);
  };
  double pbrnda_out;

  if (tkew < 0. && pmxl > mltmin * ONEM) {
    // ---- TKE deficit: reduce the mixed layer depth until the balance is restored, :388-450 -----------------------------
    if (PR(3) * lbi > 1.) {
      pmxl = 1. / lbi;
      dpmxl = fmin3(pmxl - PR(1), PR(3) - pmxl, TENCM);
      pmxl = pmxl - .5 * dpmxl;
    } else {
      dpmxl = -TENCM;
      pmxl = PR(3) + dpmxl;
    }
    tkeo = tkew;
    nitr = 0;
    const double pres1 = PR(1), pres3 = PR(3);
    for (;;) {
      nitr = nitr + 1;
      rm1 = exp_libm(-lui * pmxl);
      q = lei * fmax2(TENCM, pmxl);
      rm3 = exp_libm(-q);
      rm4 = 2. / q;
      q = (cbftot - cbfpsw * (rm4 * (1. - rm3) - rm3));
      if (q < 0.) {
        lbi = lui;
        rm2 = rm1;
      } else {
        lbi = lui * kappa / mu;
        rm2 = exp_libm(-lbi * pmxl);
      }
      mtkeus = cus * rm1;
      mtkeni = cni * rm1;
      mtkebf = q * rm2 * pmxl;
      mtkers = -crs * pmxl * pmxl * pmxl;
      mtkepe = 0.;
      mtkeke = 0.;
      tkew = mtkeus + mtkeni + mtkebf + mtkers;
      if (!(nitr == 1 && pres3 * lbi > 1.)) {
        dtke = (tkew - tkeo) / dpmxl;
        if (fabs(dtke) < (fabs(tkew) + 1.e-22) / (pres3 - pres1)) {
          if (tkew < 0.) dpmxl = .5 * (pres1 - pmxl);
          else dpmxl = .5 * (pres3 - pmxl);
        } else
          dpmxl = fmax2(pres1 - pmxl, fmin2(pres3 - pmxl, -tkew / dtke));
      }
      pmxl = pmxl + dpmxl;
      tkeo = tkew;
      if (fabs(dpmxl) < ONEMM || nitr == MAXITR) break;
    }
    // (nitr == maxitr: the reference prints the column -- 'reached maxitr when detraining', :439-440 -- and goes on; counted as it prints)
    if (nitr == MAXITR) atomicAdd(M.maxitr_count, 1);
    KPROF_MARK(wid, 2);

    pmxl = fmax2(mltmin * ONEM, pmxl);
    dpfsl = PR(3) - pmxl;
    dptopl = fmin2(thktop * ONEM, .5 * (pmxl - PR(1)));

    if (pmxl < PR(2)) {                                                               // :456-471
      q = 1. / dpfsl;
      tfsl = (TT(2) * DP(2) + TT(1) * (PR(2) - pmxl)) * q;
      sfsl = (SS(2) * DP(2) + SS(1) * (PR(2) - pmxl)) * q;
      TT(2) = TT(1);
      SS(2) = SS(1);
      {
        const double dp2 = DP(2), w1 = PR(2) - pmxl;
        tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = TR(nt, 2); y = TR(nt, 1); },
                 [&](int nt, double x, double y, double) { trfsl[nt] = (x * dp2 + y * w1) * q; TR(nt, 2) = y; });
      }
      DP(2) = pmxl - PR(1) - dptopl;
    } else {                                                                          // :473-509
      tfsl = TT(2);
      sfsl = SS(2);
      tracers4(ntr, [&](int nt, double &x, double &, double &) { x = TR(nt, 2); }, [&](int nt, double x, double, double) { trfsl[nt] = x; });
      DP(2) = pmxl - PR(2);
      if (DP(1) > dptopl) {
        dpt = DP(1) - dptopl;
        q = 1. / (DP(2) + dpt);
        TT(2) = (TT(2) * DP(2) + TT(1) * dpt) * q;
        SS(2) = (SS(2) * DP(2) + SS(1) * dpt) * q;
        {
          const double dp2 = DP(2);
          tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = TR(nt, 2); y = TR(nt, 1); },
                   [&](int nt, double x, double y, double) { TR(nt, 2) = (x * dp2 + y * dpt) * q; });
        }
        DP(2) = DP(2) + dpt;
      } else {
        dpt = dptopl - DP(1);
        q = 1. / (DP(1) + dpt);
        TT(1) = (TT(1) * DP(1) + TT(2) * dpt) * q;
        SS(1) = (SS(1) * DP(1) + SS(2) * dpt) * q;
        {
          const double dp1 = DP(1);
          tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = TR(nt, 1); y = TR(nt, 2); },
                   [&](int nt, double x, double y, double) { TR(nt, 1) = (x * dp1 + y * dpt) * q; });
        }
        DP(2) = DP(2) - dpt;
      }
    }
    DP(1) = dptopl;

    // ---- forcing, :517-664 ------------------------------------------------------------------------------------------------
    kmax = 1;
    for (int k0 = 2; k0 <= kk; k0 += COLUMN_U) {
      double a[COLUMN_U];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) a[u] = DP(k0 + u <= kk ? k0 + u : kk);
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++)
        if (k0 + u <= kk && a[u] > EPSILP) kmax = k0 + u;
    }
    kfmax = 0;

    // brine forcing below the surface layer
    pbrnda_out = 0.;
    if (brnflx < 0.) {
      if (kfpl > kmax) {
        if (dpfsl > ONEMU) {
          bpmldp = fmin2(bpmndp, dpfsl + DP(2));
          q = brnflx * delt1 * GRAV / bpmldp;
          SS(2) = SS(2) - q * fmax2(0., bpmldp - dpfsl) / DP(2);
          sfsl = sfsl - q * fmin2(dpfsl, bpmldp) / dpfsl;
        } else
          SS(2) = SS(2) - brnflx * delt1 * GRAV / DP(2);
      } else {
        pup = PR(3);
        drhup = 0.;
        k = kfpl;
        while (k <= kmax) {
          if (DP(k) > ONEMU) {
            plo = PR(k) + .5 * DP(k);
            drhlo = eos::rho(plo, TT(k), SS(k)) - eos::rho(plo, TT(1), SS(1));
            if (drhlo > bpdrho) break;
            pup = plo;
            drhup = drhlo;
          }
          k = k + 1;
        }
        if (k > kmax) pbrnd = PR(kmax + 1);
        else pbrnd = ((drhlo - bpdrho) * pup + (bpdrho - drhup) * plo) / (drhlo - drhup);
        pbrnd = fmin2(pbrnd, PR(3) + bpmxdp);
        pbrnda_out = pbrnd;
        k = kfpl;
        bcwsum = 0.;
        bdpsum = 0.;
        tup = tfsl;
        sup = sfsl;
        dup = SIG(tfsl, sfsl);
        while (k < kmax && PR(k + 1) < pbrnd) {
          if (k == kfpl || dup < DR(k)) {
            dsgdt = eosd::dsigdt(P, TT(k), SS(k));
            dsgds = eosd::dsigds(P, TT(k), SS(k));
            const double b = fmax2(dsgmnr * (DR(k) - DR(k - 1)), dsgdt * (TT(k) - tup) + dsgds * (SS(k) - sup)) / (dsgds * fmax2(bpdpmn, DP(k)));
            BC(k) = b;
            bcwsum = bcwsum + b * DP(k);
            bdpsum = bdpsum + DP(k);
          } else
            BC(k) = 0.;
          tup = TT(k);
          sup = SS(k);
          dup = DN(k);
          k = k + 1;
        }
        if (k == kfpl || dup < DR(k)) {
          dsgdt = eosd::dsigdt(P, TT(k), SS(k));
          dsgds = eosd::dsigds(P, TT(k), SS(k));
          const double dd = fmax2(bpdpmn, DP(k));
          const double b = fmax2(dsgmnr * (DR(k) - DR(k - 1)), dsgdt * (TT(k) - tup) + dsgds * (SS(k) - sup)) * fmax2(bpdpmn, pbrnd - PR(k)) /
                           (dsgds * (dd * dd));
          BC(k) = b;
          bcwsum = bcwsum + b * DP(k);
          bdpsum = bdpsum + DP(k);
        } else
          BC(k) = 0.;
        kfmax = k;
        if (bdpsum <= EPSILP) {
          if (dpfsl > ONEMU) {
            bpmldp = fmin2(bpmndp, dpfsl + DP(2));
            q = brnflx * delt1 * GRAV / bpmldp;
            SS(2) = SS(2) - q * fmax2(0., bpmldp - dpfsl) / DP(2);
            sfsl = sfsl - q * fmin2(dpfsl, bpmldp) / dpfsl;
          } else
            SS(2) = SS(2) - brnflx * delt1 * GRAV / DP(2);
        } else {
          if (bdpsum < bpmndp) {
            bpmldp = fmin2(bpmndp, bdpsum + dpfsl + DP(2));
            q = brnflx * delt1 * GRAV / bpmldp;
            SS(2) = SS(2) - q * fmax2(0., bpmldp - bdpsum - dpfsl) / DP(2);
            if (dpfsl > ONEMU) {
              sfsl = sfsl - q * fmin2(dpfsl, bpmldp - bdpsum) / dpfsl;
              bpc = q * bdpsum / bcwsum;
            } else
              bpc = q * (bdpsum + dpfsl) / bcwsum;
          } else
            bpc = brnflx * delt1 * GRAV / bcwsum;
          for (k = kfpl; k <= kfmax; k++) SS(k) = SS(k) - bpc * BC(k);
        }
      }
    }

    // heat forcing below the surface layer, :626-650
    pswbas = swfc2 * exp_libm(-lei * DP(1));
    pswup = pswbas;
    pswlo = swfc2 * exp_libm(-lei * fmin2(pradd, pmxl));
    q = delt1 * GRAV / DP(2);
    TT(2) = TT(2) - (pswup - pswlo) * sswflx * q / SPCIFH;
    pswup = pswlo;
    pswlo = swfc2 * exp_libm(-lei * fmin2(pradd, PR(3)));
    if (dpfsl > ONEMU) {
      tfsl = tfsl - (pswup - pswlo) * sswflx * delt1 * GRAV / (SPCIFH * dpfsl);
      pswup = pswlo;
    }
    k = kfpl;
    while (k < kmax) {
      if (DP(k) > ONEMU) {
        pswlo = swfc2 * exp_libm(-lei * fmin2(pradd, PR(k + 1)));
        TT(k) = TT(k) - (pswup - pswlo) * sswflx * delt1 * GRAV / (SPCIFH * DP(k));
        pswup = pswlo;
        kfmax = kfmax > k ? kfmax : k;
      }
      k = k + 1;
      if (PR(k) > pradd) break;
    }

    // heat and salt forcing of the top layer, :652-664
    q = delt1 * GRAV / DP(1);
    TT(1) = TT(1) - (surflx - (pswbas - pswup) * sswflx + surrlx) * q / SPCIFH;
    SS(1) = SS(1) - (salflx - brnflx + salrlx) * q;
    tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = TR(nt, 1); y = V.f[F_trflx][c + (size_t)nt * np]; },
             [&](int nt, double x, double y, double) { TR(nt, 1) = x - y * q; });

    // density of the layers the forcing touched, :666-671
    DN(1) = SIG(TT(1), SS(1));
    DN(2) = SIG(TT(2), SS(2));
    for (k = kfpl; k <= kfmax; k++) DN(k) = SIG(TT(k), SS(k));

    if (dpfsl <= ONEMU) {                                                             // :673-683
      q = 1. / (dpfsl + DP(2));
      TT(2) = (tfsl * dpfsl + TT(2) * DP(2)) * q;
      SS(2) = (sfsl * dpfsl + SS(2) * DP(2)) * q;
      DN(2) = SIG(TT(2), SS(2));
      {
        const double dp2 = DP(2);
        tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = trfsl[nt]; y = TR(nt, 2); },
                 [&](int nt, double x, double y, double) { TR(nt, 2) = (x * dpfsl + y * dp2) * q; });
      }
      DP(2) = dpfsl + DP(2);
    } else {
      // ---- the fossil mixed layer goes into isopycnic layers, :685-805 ------------------------------------------------
#define MIX_FSL(kq, dd)                                                                             \
  {                                                                                                \
    q = 1. / ((dd) + DP(kq));                                                                       \
    TT(kq) = (tfsl * (dd) + TT(kq) * DP(kq)) * q;                                                   \
    SS(kq) = (sfsl * (dd) + SS(kq) * DP(kq)) * q;                                                   \
    DN(kq) = SIG(TT(kq), SS(kq));                                                                   \
    {                                                                                              \
      const double dd_ = (dd), dpk_ = DP(kq);                                                      \
      tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = trfsl[nt]; y = TR(nt, kq); }, \
               [&](int nt, double x, double y, double) { TR(nt, kq) = (x * dd_ + y * dpk_) * q; }); \
    }                                                                                              \
    DP(kq) = (dd) + DP(kq);                                                                         \
  }
      k = kk < kfpl ? kk : kfpl;
      if (k == 3) {
        MIX_FSL(k, dpfsl)
      } else {
        q = 1. / (dpfsl + DP(k));
        ttmp = (tfsl * dpfsl + TT(k) * DP(k)) * q;
        stmp = (sfsl * dpfsl + SS(k) * DP(k)) * q;
        sigtmp = SIG(ttmp, stmp);
        sigfsl = SIG(tfsl, sfsl);
        if (sigtmp >= DR(k)) {
          if (sigfsl > DN(k) && DN(k) <= DN(kk < k + 1 ? kk : k + 1) && eos::rho(pmxl, tfsl, sfsl) < eos::rho(pmxl, TT(k), SS(k))) {
            k = k - 1;
            MIX_FSL(k, dpfsl)
          } else {
            TT(k) = ttmp;
            SS(k) = stmp;
            DN(k) = sigtmp;
            {
              const double dpk_ = DP(k);
              tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = trfsl[nt]; y = TR(nt, k); },
                       [&](int nt, double x, double y, double) { TR(nt, k) = (x * dpfsl + y * dpk_) * q; });
            }
            DP(k) = dpfsl + DP(k);
          }
        } else {
          if (DP(k) > ONEMU && DN(k) > DR(k) && sigfsl < DR(k) - 1.e-6) {
            dps = fmin2(dpfsl, DP(k) * (DN(k) - DR(k)) / (DR(k) - sigfsl));
            MIX_FSL(k, dps)
            dpfsl = dpfsl - dps;
            if (dpfsl <= ONEMU) {
              MIX_FSL(2, dpfsl)
            } else {
              k = k - 1;
              while (sigfsl < DR(k)) {
                if (k == 3) break;
                k = k - 1;
              }
              MIX_FSL(k, dpfsl)
            }
          } else {
            k = k - 1;
            while (sigfsl < DR(k)) {
              if (k == 3) break;
              k = k - 1;
            }
            MIX_FSL(k, dpfsl)
          }
        }
      }
    }

    KPROF_MARK(wid, 3);
  } else {

    if (tkew < 0.) {                                                                  // :811-834
      pmxl = mltmin * ONEM;
      tdps = TT(2) * DP(2);
      sdps = SS(2) * DP(2);
      {
        const double dp2 = DP(2);
        trd_init(dp2);
      }
      k = kfpl;
      while (k <= kk) {
        q = fmin2(pmxl, PR(k + 1)) - PR(k);
        tdps = tdps + TT(k) * q;
        sdps = sdps + SS(k) * q;
        trd_add(k, q);
        DP(k) = PR(k + 1) - fmin2(pmxl, PR(k + 1));
        if (PR(k + 1) > pmxl) break;
        k = k + 1;
      }
    } else {
      // ---- TKE surplus: deepen the mixed layer until the balance is restored, :836-1018 ---------------------------------
      const double pres1 = PR(1);
      q = 1. / (DP(1) + DP(2));
      tmxl0 = (TT(1) * DP(1) + TT(2) * DP(2)) * q;
      smxl0 = (SS(1) * DP(1) + SS(2) * DP(2)) * q;
#define UU(k, d) uu[(size_t)(k) * np + (d)]
#define VV(k, d) vv[(size_t)(k) * np + (d)]
#define DPU(k, d) dpu[(size_t)(k) * np + (d)]
#define DPV(k, d) dpv[(size_t)(k) * np + (d)]
      um = (UU(1, 0) * DPU(1, 0) + UU(1, 1) * DPU(1, 1) + UU(2, 0) * DPU(2, 0) + UU(2, 1) * DPU(2, 1)) /
           fmax2(ONECM, DPU(1, 0) + DPU(1, 1) + DPU(2, 0) + DPU(2, 1));
      vm = (VV(1, 0) * DPV(1, 0) + VV(1, ni) * DPV(1, ni) + VV(2, 0) * DPV(2, 0) + VV(2, ni) * DPV(2, ni)) /
           fmax2(ONECM, DPV(1, 0) + DPV(1, ni) + DPV(2, 0) + DPV(2, ni));
      dpe0 = 0.;
      dke0 = 0.;
      tkeu = tkew;
      k = kfpl;
      tdps = TT(2) * DP(2);
      sdps = SS(2) * DP(2);
      {
        const double dp2 = DP(2);
        trd_init(dp2);
      }
      // One loop, not two nested ones: a wavefront walks a loop nest in step -- while one lane iterates on its partly entrained layer (up to
      // maxitr trips of ~2 000 instructions, three exp() and four evaluations of the equation of state) the 63 others, done with theirs after
      // one trip, wait, and at the next layer another lane makes them wait again: the wave's trips were the SUM over the layers of the slowest
      // lane's iterations (71 trips of 3.8 us in the slowest waves after 1 000 steps, measured per wave: make kprof).  Here a lane carries its
      // own position -- st 0: at layer k, not entered; 1: iterating on layer k; 2: done -- and a trip does, for every lane, whatever that lane
      // does next: the wave's trips are the LONGEST lane's own sequence.  A lane's arithmetic and its order are unchanged.
      // A trip: (1) the loads of the layer the lane may enter next are requested -- k + 1 while it iterates on k, k itself before it has
      // entered one; (2) one iteration of the lanes that are inside a layer, and when it was the last what follows it; (3) the lanes in front
      // of a layer enter it, or step over it when it has no mass, with the values requested in (1): their latency lies under (2).
      {
        int st = 0;
        double presk = 0., presk1 = 0., tk = 0., sk = 0., delpk = 0.;
        while (st != 2) {
          const int kq = (st == 1 ? k + 1 : k) <= kk ? (st == 1 ? k + 1 : k) : kk;
          const double l_dp = DP(kq), l_p0 = PR(kq), l_p1 = PR(kq + 1), l_t = TT(kq), l_s = SS(kq);
          const double l_u0 = UU(kq, 0), l_u1 = UU(kq, 1), l_du0 = DPU(kq, 0), l_du1 = DPU(kq, 1);
          const double l_v0 = VV(kq, 0), l_v1 = VV(kq, ni), l_dv0 = DPV(kq, 0), l_dv1 = DPV(kq, ni);
          double l_tr[4];                // the tracers of the layer being iterated on (read when the iteration ends in this trip)
          {
            const int kt = k <= kk ? k : kk;
#pragma unroll
            for (int b = 0; b < 4; b++) l_tr[b] = TR(b < ntr ? b : ntr - 1, kt);
          }
          if (st == 1) {
            bool fin = false;
            nitr = nitr + 1;
            KPROF_TRIP(wid, 7);
            tmxl = (tmxl0 * (presk - pres1) + tk * (pmxl - presk)) / (pmxl - pres1);
            smxl = (smxl0 * (presk - pres1) + sk * (pmxl - presk)) / (pmxl - pres1);
            dpe = dpe0 + fmax2(.5 * ALPHA0 * ALPHA0 * mldjmp * (presk - pres1) * (pmxl - presk),
                               eos0::p_p_alpha(pmxl, pres1, tmxl, smxl) - eos0::p_p_alpha(pmxl, presk, tk, sk) -
                                   eos0::p_p_alpha(presk, pres1, tmxl0, smxl0) - (pres1 - presk) * eos::p_alpha(pmxl, presk, tk, sk)) *
                             ALPHA0 / (delt1 * GRAV);
            dke = dke0 + .5 * rm5 * (presk - pres1) * (pmxl - presk) * ((uk - um) * (uk - um) + (vk - vm) * (vk - vm)) * ALPHA0 /
                             ((pmxl - pres1) * delt1 * GRAV);
            rm1 = exp_libm(-lui * pmxl);
            q = lei * pmxl;
            rm3 = exp_libm(-q);
            rm4 = 2. / q;
            q = (cbftot - cbfpsw * (rm4 * (1. - rm3) - rm3));
            if (q < 0.) {
              lbi = lui;
              rm2 = rm1;
            } else {
              lbi = lui * kappa / mu;
              rm2 = exp_libm(-lbi * pmxl);
            }
            mtkeus = cus * rm1;
            mtkeni = cni * rm1;
            mtkebf = q * rm2 * pmxl;
            mtkers = -crs * pmxl * pmxl * pmxl;
            mtkepe = -dpe;
            mtkeke = dke;
            tkew = mtkeus + mtkeni + mtkebf + mtkers + mtkepe + mtkeke;
            if (nitr == 1 && tkew > 0.) fin = true;
            else {
              if (nitr == 1) {
                pmxl = presk;
                dpmxl = fmin2(TENCM, .5 * delpk);
                tkel = tkew;
                tkew = tkeu;
              } else {
                dtke = (tkew - tkeo) / dpmxl;
                bool chngd = false;
                if (nitr == 2) {
                  if (dtke > -tkew / (presk1 - pmxl)) {
                    pmxl = presk1;
                    dpmxl = -fmin2(TENCM, .5 * delpk);
                    tkew = tkel;
                    chngd = true;
                  }
                }
                if (!chngd) {
                  if (fabs(dtke) < (fabs(tkew) + 1.e-22) / delpk) {
                    if (tkew < 0.) dpmxl = .5 * (presk - pmxl);
                    else dpmxl = presk1 - pmxl;
                  } else
                    dpmxl = fmax2(presk - pmxl, fmin2(presk1 - pmxl, -tkew / dtke));
                  dpmxl = fmax2(fmax2(mltmin * ONEM, presk) - pmxl, dpmxl);
                }
              }
              pmxl = pmxl + dpmxl;
              tkeo = tkew;
              if (fabs(dpmxl) < ONEMM || nitr == MAXITR) fin = true;
            }
            if (fin) {
              // (nitr == maxitr: the reference prints the column -- 'reached maxitr when entraining', :949-950 -- and goes on; counted as it prints)
              if (nitr == MAXITR) atomicAdd(M.maxitr_count + 1, 1);
              if (pmxl < presk1 - EPSILP && nitr < MAXITR) {
                tdps = tdps + tk * (pmxl - presk);
                sdps = sdps + sk * (pmxl - presk);
                {
                  const double w_ = pmxl - presk;
                  trd_add_pre(k, w_, l_tr);
                }
                DP(k) = presk1 - pmxl;
                st = 2;
              } else {
                tdps = tdps + tk * delpk;
                sdps = sdps + sk * delpk;
                trd_add_pre(k, delpk, l_tr);
                // :996-1009: the mixed layer's properties, potential and kinetic energy change with the whole layer entrained.  When the
                // iteration ended with its first evaluation (the common case: TKE left after the whole layer) that evaluation WAS at
                // pmxl = pres(k+1), with these very expressions on these very operands: tmxl, smxl, dpe, dke hold the values already
                if (nitr != 1) {
                  pmxl = presk1;
                  tmxl = (tmxl0 * (presk - pres1) + tk * (pmxl - presk)) / (pmxl - pres1);
                  smxl = (smxl0 * (presk - pres1) + sk * (pmxl - presk)) / (pmxl - pres1);
                  dpe = dpe0 + fmax2(.5 * ALPHA0 * ALPHA0 * mldjmp * (presk - pres1) * (pmxl - presk),
                                     eos0::p_p_alpha(pmxl, pres1, tmxl, smxl) - eos0::p_p_alpha(pmxl, presk, tk, sk) -
                                         eos0::p_p_alpha(presk, pres1, tmxl0, smxl0) - (pres1 - presk) * eos::p_alpha(pmxl, presk, tk, sk)) *
                                   ALPHA0 / (delt1 * GRAV);
                  dke = dke0 + .5 * rm5 * (presk - pres1) * (pmxl - presk) * ((uk - um) * (uk - um) + (vk - vm) * (vk - vm)) * ALPHA0 /
                                   ((pmxl - pres1) * delt1 * GRAV);
                }
                dpe0 = dpe;
                dke0 = dke;
                tmxl0 = tmxl;
                smxl0 = smxl;
                um = (um * (presk - pres1) + uk * (pmxl - presk)) / (pmxl - pres1);
                vm = (vm * (presk - pres1) + vk * (pmxl - presk)) / (pmxl - pres1);
                DP(k) = 0.;
                k = k + 1;
                st = 0;
              }
            }
          }
          if (st == 0) {                 // (the values requested at the top of the trip are layer k's: k + 1 of the layer just left, or k itself)
            if (k > kk) st = 2;
            else if (l_dp < EPSILP) k = k + 1;
            else {
              presk = l_p0; presk1 = l_p1; tk = l_t; sk = l_s; delpk = l_dp;
              pmxl = presk1;
              uk = (l_u0 * l_du0 + l_u1 * l_du1) / fmax2(ONECM, l_du0 + l_du1);
              vk = (l_v0 * l_dv0 + l_v1 * l_dv1) / fmax2(ONECM, l_dv0 + l_dv1);
              nitr = 0;
              st = 1;
            }
          }
        }
      }
    }

    KPROF_MARK(wid, 4);
    {                                                                                 // :1020-1034
      const double p3 = fmin2(PR(kk + 1), pmxl);
      PR(3) = p3;
      DP(2) = p3 - PR(2);
      q = 1. / DP(2);
      TT(2) = tdps * q;
      SS(2) = sdps * q;
#pragma unroll
      for (int b = 0; b < 4; b++)
        if (b < ntr) TR(b, 2) = trd4[b] * q;
      tracers4(ntr, [&](int nt, double &x, double &, double &) { x = trdps[nt]; }, [&](int nt, double x, double, double) { TR(nt, 2) = x * q; }, 4);
      kfpl = k;
      for (k = 4; k <= kfpl; k++) PR(k) = p3;
    }

    // top layer back to its reference thickness, :1036-1063
    dptopl = fmin2(thktop * ONEM, .5 * (PR(3) - PR(1)));
    if (DP(1) > dptopl) {
      dpt = DP(1) - dptopl;
      q = 1. / (DP(2) + dpt);
      TT(2) = (TT(2) * DP(2) + TT(1) * dpt) * q;
      SS(2) = (SS(2) * DP(2) + SS(1) * dpt) * q;
      {
        const double dp2 = DP(2);
        tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = TR(nt, 2); y = TR(nt, 1); },
                 [&](int nt, double x, double y, double) { TR(nt, 2) = (x * dp2 + y * dpt) * q; });
      }
      DP(2) = DP(2) + dpt;
    } else {
      dpt = dptopl - DP(1);
      q = 1. / (DP(1) + dpt);
      TT(1) = (TT(1) * DP(1) + TT(2) * dpt) * q;
      SS(1) = (SS(1) * DP(1) + SS(2) * dpt) * q;
      {
        const double dp1 = DP(1);
        tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = TR(nt, 1); y = TR(nt, 2); },
                 [&](int nt, double x, double y, double) { TR(nt, 1) = (x * dp1 + y * dpt) * q; });
      }
      DP(2) = DP(2) - dpt;
    }
    DP(1) = dptopl;
    PR(2) = PR(1) + DP(1);

    // ---- forcing, :1065-1196 ----------------------------------------------------------------------------------------------
    kmax = 1;
    for (int k0 = 2; k0 <= kk; k0 += COLUMN_U) {
      double a[COLUMN_U];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) a[u] = DP(k0 + u <= kk ? k0 + u : kk);
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++)
        if (k0 + u <= kk && a[u] > EPSILP) kmax = k0 + u;
    }
    kfmax = 0;

    pbrnda_out = 0.;
    if (brnflx < 0.) {
      if (kfpl > kmax) SS(2) = SS(2) - brnflx * delt1 * GRAV / DP(2);
      else {
        pup = PR(3);
        drhup = 0.;
        k = kfpl;
        while (k <= kmax) {
          if (DP(k) > ONEMU) {
            plo = PR(k) + .5 * DP(k);
            drhlo = eos::rho(plo, TT(k), SS(k)) - eos::rho(plo, TT(1), SS(1));
            if (drhlo > bpdrho) break;
            pup = plo;
            drhup = drhlo;
          }
          k = k + 1;
        }
        if (k > kmax) pbrnd = PR(kmax + 1);
        else pbrnd = ((drhlo - bpdrho) * pup + (bpdrho - drhup) * plo) / (drhlo - drhup);
        pbrnd = fmin2(pbrnd, PR(3) + bpmxdp);
        pbrnda_out = pbrnd;
        k = kfpl;
        bcwsum = 0.;
        bdpsum = 0.;
        tup = TT(2);
        sup = SS(2);
        dup = SIG(TT(2), SS(2));
        while (k < kmax && PR(k + 1) < pbrnd) {
          if (k == kfpl || dup < DR(k)) {
            dsgdt = eosd::dsigdt(P, TT(k), SS(k));
            dsgds = eosd::dsigds(P, TT(k), SS(k));
            const double b = fmax2(dsgmnr * (DR(k) - DR(k - 1)), dsgdt * (TT(k) - tup) + dsgds * (SS(k) - sup)) / (dsgds * fmax2(bpdpmn, DP(k)));
            BC(k) = b;
            bcwsum = bcwsum + b * DP(k);
            bdpsum = bdpsum + DP(k);
          } else
            BC(k) = 0.;
          tup = TT(k);
          sup = SS(k);
          dup = DN(k);
          k = k + 1;
        }
        if (k == kfpl || dup < DR(k)) {
          dsgdt = eosd::dsigdt(P, TT(k), SS(k));
          dsgds = eosd::dsigds(P, TT(k), SS(k));
          const double dd = fmax2(bpdpmn, DP(k));
          const double b = fmax2(dsgmnr * (DR(k) - DR(k - 1)), dsgdt * (TT(k) - tup) + dsgds * (SS(k) - sup)) * fmax2(bpdpmn, pbrnd - PR(k)) /
                           (dsgds * (dd * dd));
          BC(k) = b;
          bcwsum = bcwsum + b * DP(k);
          bdpsum = bdpsum + DP(k);
        } else
          BC(k) = 0.;
        kfmax = k;
        if (bdpsum <= EPSILP) SS(2) = SS(2) - brnflx * delt1 * GRAV / DP(2);
        else {
          if (bdpsum < bpmndp) {
            bpmldp = fmin2(bpmndp, bdpsum + DP(2));
            q = brnflx * delt1 * GRAV / bpmldp;
            SS(2) = SS(2) - q * (bpmldp - bdpsum) / DP(2);
            bpc = q * bdpsum / bcwsum;
          } else
            bpc = brnflx * delt1 * GRAV / bcwsum;
          for (k = kfpl; k <= kfmax; k++) SS(k) = SS(k) - bpc * BC(k);
        }
      }
    }

    // heat forcing below the surface layer, :1162-1180
    pswbas = swfc2 * exp_libm(-lei * DP(1));
    pswup = pswbas;
    pswlo = swfc2 * exp_libm(-lei * fmin2(pradd, PR(3)));
    q = delt1 * GRAV / DP(2);
    TT(2) = TT(2) - (pswup - pswlo) * sswflx * q / SPCIFH;
    pswup = pswlo;
    k = kfpl;
    while (k < kmax) {
      if (DP(k) > ONEMU) {
        pswlo = swfc2 * exp_libm(-lei * fmin2(pradd, PR(k + 1)));
        TT(k) = TT(k) - (pswup - pswlo) * sswflx * delt1 * GRAV / (SPCIFH * DP(k));
        pswup = pswlo;
        kfmax = kfmax > k ? kfmax : k;
      }
      k = k + 1;
      if (PR(k) > pradd) break;
    }

    // heat and salt forcing of the top layer, :1182-1190
    q = delt1 * GRAV / DP(1);
    TT(1) = TT(1) - (surflx - (pswbas - pswup) * sswflx + surrlx) * q / SPCIFH;
    SS(1) = SS(1) - (salflx - brnflx + salrlx) * q;
    tracers4(ntr, [&](int nt, double &x, double &y, double &) { x = TR(nt, 1); y = V.f[F_trflx][c + (size_t)nt * np]; },
             [&](int nt, double x, double y, double) { TR(nt, 1) = x - y * q; });

    DN(1) = SIG(TT(1), SS(1));                                                        // :1192-1196
    DN(2) = SIG(TT(2), SS(2));
    for (k = kfpl; k <= kfmax; k++) DN(k) = SIG(TT(k), SS(k));
    KPROF_MARK(wid, 5);
  }

  V.f[F_mtkeus][c] = mtkeus;
  V.f[F_mtkeni][c] = mtkeni;
  V.f[F_mtkebf][c] = mtkebf;
  V.f[F_mtkers][c] = mtkers;
  V.f[F_mtkepe][c] = mtkepe;
  V.f[F_mtkeke][c] = mtkeke;
  V.f[F_pbrnda][c] = pbrnda_out;

  // first physical layer, :1200-1214
  k = 3;
  dps = 0.;
  {
    bool walking = true;                               // (COLUMN_U levels' loads ahead, as in k_convec_column)
    for (int k0 = 3; walking && k0 <= kk; k0 += COLUMN_U) {
      double a[COLUMN_U];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) a[u] = DP(k0 + u <= kk ? k0 + u : kk);
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) {
        if (!walking || k0 + u > kk) break;
        if (a[u] < EPSILP) {
          dps = dps + a[u];
          DP(k0 + u) = 0.;
          k = k0 + u + 1;
        } else
          walking = false;
      }
    }
  }
  if (k > kk) DP(2) = DP(2) + dps;
  else DP(k) = DP(k) + dps;
  V.m[I_kfpla][c + (size_t)(n - 1) * np] = k;
  KPROF_MARK(wid, 6);
}

// ---- the copy-back rules, :1216-1241: negative salinities and tracers are set to zero and what that adds is booked in salt_corr,
//      trc_corr; the turbulence tracers are bounded below.  (For non-negative values the reference's updates subtract zero, and
//      the sums they are subtracted from are never -0: nothing is done for them.)  One thread per column AND array (blockIdx.y = 0
//      salinity, 1.. the tracers): the columns of the ntr + 1 arrays are independent of each other, so they go to as many
//      wavefronts instead of being walked one after the other by the column's one thread.
__global__ __launch_bounds__(64) void k_mxl_clamp(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int kk = V.kk;
  const size_t np = V.nplane;
  const Params &P = V.P;
  gcd_t delp = V.f[F_dp] + c + (size_t)nn * np - np;
  const int nt = (int)by_ - 1;
  gd_t x = (nt < 0 ? V.f[F_saln] : V.f[F_trc] + (size_t)nt * 2 * kk * np) + c + (size_t)nn * np - np;
#define XX(k) x[(size_t)(k) * np]
  const bool is_tke = nt >= 0 && P.itrtke >= 1 && nt + 1 == P.itrtke, is_gls = nt >= 0 && P.itrtke >= 1 && P.gls && nt + 1 == P.itrgls;
  if (is_tke || is_gls) {
    const double lo = is_tke ? TKE_MIN : GLS_PSI_MIN;
    for (int k0 = 1; k0 <= kk; k0 += 2 * COLUMN_U) {
      double a[2 * COLUMN_U];
#pragma unroll
      for (int u = 0; u < 2 * COLUMN_U; u++) a[u] = XX(k0 + u <= kk ? k0 + u : kk);
#pragma unroll
      for (int u = 0; u < 2 * COLUMN_U; u++)
        if (k0 + u <= kk && !(a[u] > lo)) XX(k0 + u) = fmax2(a[u], lo);
    }
    return;
  }
  gd_t corr = nt < 0 ? V.f[F_salt_corr] + c : V.f[F_trc_corr] + c + (size_t)nt * np;
  double tc = 0.;
  bool any = false;
  for (int k0 = 1; k0 <= kk; k0 += 2 * COLUMN_U) {
    double a[2 * COLUMN_U];
#pragma unroll
    for (int u = 0; u < 2 * COLUMN_U; u++) a[u] = XX(k0 + u <= kk ? k0 + u : kk);
#pragma unroll
    for (int u = 0; u < 2 * COLUMN_U; u++)
      if (k0 + u <= kk && a[u] < 0.) {
        if (!any) { tc = *corr; any = true; }
        tc = tc - fmin2(0., a[u]) * delp[(size_t)(k0 + u) * np] / GRAV;
        XX(k0 + u) = 0.;
      }
  }
  if (any) *corr = tc;
#undef XX
}

int st_mxlayr(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)mm; (void)k1m;
  const DevView &h = c->h;
  if (h.P.vcoord_tag != 1) return ctx_fail(c, "mxlayr is only called for isopyc_bulkml (phy/mod_blom_step.F90:188-192)");
  if (h.ntr > MAXTR_MXL) return ctx_fail(c, "mxlayr: too many tracers for the device kernel");
  MxlPar M;
  if (c->mlrttp == "variable") M.rtsflg = 1;
  else if (c->mlrttp == "constant") M.rtsflg = 2;
  else if (c->mlrttp == "limited") M.rtsflg = 3;
  else return ctx_fail(c, " mlrttp = " + c->mlrttp + " is unsupported!");                       // :203-212
  M.rm0 = c->rm0; M.rm5 = c->rm5; M.ce = c->eddtra_ce; M.rtau = 1. / c->tau_mlr; M.rlf = 1. / c->lfmin;
  M.niwgf = c->niwgf; M.niwbf = c->niwbf; M.swamxd = c->swamxd; M.mltmin = 5.; M.thktop = 10.;
  if (int rc = ctx_err_words(c)) return rc;
  M.maxitr_count = c->err_dev + 5;          // words 5, 6: counters, not errors
  TimeScope ts(c, "mxlayr");
  hipLaunchKernelGGL(k_mxl_bg2_sig, plane_grid(h), dim3(256), 0, c->stream, c->d, nn);
  if (int rc = st_xctilr(c, h.f[F_util1], 1, 1, 1, 1, 1)) return rc;
  hipLaunchKernelGGL(k_mxl_bg2_grad, plane_grid(h), dim3(256), 0, c->stream, c->d);
  hipLaunchKernelGGL(k_mxl_bg2_sum, plane_grid(h), dim3(256), 0, c->stream, c->d);
  if (c->diapfl_mom_on_side) {               // diapfl's momentum mixing on the second stream (stage_diapfl.hip): u, v, dpu, dpv are read from here on
    c->diapfl_mom_on_side = false;
    if (int rc = ctx_side_join(c, 7)) return rc;
  }
  if (c->rm5 > 0.) {                                                                            // :287-290
    if (int rc = st_xctilr(c, h.f[F_u] + (size_t)(k1n - 1) * h.nplane, 1, h.kk, 1, 1, 13)) return rc;
    if (int rc = st_xctilr(c, h.f[F_v] + (size_t)(k1n - 1) * h.nplane, 1, h.kk, 1, 1, 14)) return rc;
  }
  hipLaunchKernelGGL(k_mxl_column, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, M, n, nn KPROF_PASS(3));
  // The copy-back clamp reads dp at the interior points (the halo update below writes halo points) and writes S, the tracers and their correction
  // planes there; what follows in this stage -- pu, pv, p, dpu, dpv, the velocities onto the new layers -- and barotp read none of them: inside
  // blomgpu_step (phys_dag bit 8) it runs on the second stream, updtrc's kernel behind it; blomgpu_step waits in front of pbcor2.  Not with the
  // arctic patch: there the halo update also rewrites the seam row j = jj, which this kernel reads.
  const bool clamp_aside = ctx_overlap_on(c) && (c->phys_dag & 8) && !c->tiling.multi() && h.nreg != 2;
  if (clamp_aside)
    if (int rc = ctx_side_fork(c, 6)) return rc;
  hipLaunchKernelGGL(k_mxl_clamp, plane_grid(h, h.ntr + 1, 64), dim3(64), 0, clamp_aside ? c->side : c->stream, c->d, nn);
  HIPCHK(c, hipGetLastError());
  if (clamp_aside) c->updtrc_on_side = true;
  // 'old' interface pressures at the velocity points (:1243-1262), the dp halo, p and the new dpu, dpv (:1264-1310), the
  // velocities onto the new layers (:1312-1374)
  if (int rc = st_mom_pupv(c, nn, 1, 0)) return rc;
  if (int rc = st_mxlayr_tail(c, nn, k1n)) return rc;
  return st_convec_velocity(c, nn);
}
