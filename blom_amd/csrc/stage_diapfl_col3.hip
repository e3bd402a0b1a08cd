// diapfl column pass with the loads of DU levels in flight: k_diapfl_column3 (stage_diapfl_col2.hip) with every sweep
// over the levels restructured like column_scan (blomgpu_internal.h).  A column's ~12 sweeps over ~40 levels each paid
// one full memory latency per level (1.0 ms for 1700 wavefronts, whatever the bandwidth); here a sweep issues the loads
// of DU levels before it uses the first.  A level's inputs never depend on what the same sweep writes to other levels
// (the recurrences travel in registers), so reading ahead changes nothing; arithmetic and order are those of
// k_diapfl_column3, whose description follows.
//
// diapfl column pass, traffic-lean form of k_diapfl_column (stage_diapfl.hip), same arithmetic.
// phy/mod_diapfl.F90:105-700.
//
// The first version mirrored the reference's ~22 private 1-D arrays as work-space planes
// (~90 F of HBM traffic per launch, F = one 3-D field).  This version
//   * works in place on temp/saln/dp/sigma/trc/difdia instead of copying them in and out.  The
//     reference moves the two mixed-layer layers to positions kmin,kmin+1 of its work arrays
//     (:159-172) and back (:580-599); here position p maps to layer 1 (p = kmin), 2 (p = kmin+1)
//     or p, and the two mixed-layer fluxes are scalars;
//   * keeps every value that the next iteration of a k-recurrence consumes in a register
//     (f0, f, gtd, 1/dsgu, 1/dsgl, fmax, fcl, fcu of the neighbouring level) -- f0 is never stored;
//   * stores only dsgu and dsgl of the six stratification arrays (dsghm, dsg, dsgui, dsgli are
//     re-derived by the reference's own expressions) and the solve's constants r = 4 cc nu dsg
//     (dsgui+dsgli), t = dsghm/4 once per level;
//   * writes fpu/fpl straight into fpug/fplg, the interface pressure into p while dp is final.
// Work planes: dsgu dsgl fcu fcl fmax h r t f ft gtd (11) + pres (wkp1).
#include "diapfl_common.h"

#define GRAV DIAPFL_GRAV
#define ALPHA0 DIAPFL_ALPHA0
#define EPSILP DIAPFL_EPSILP
#define ONEM DIAPFL_ONEM
#define MAXTR 4
// level index clamped into [lo, hi]: a look-ahead past the end of a sweep reads the last level again
#define CLAMPK(k, lo, hi) ((k) < (lo) ? (lo) : ((k) > (hi) ? (hi) : (k)))

enum { E_SU, E_SL, E_FCU, E_FCL, E_FMAX, E_H, E_R, E_T, E_F, E_FT, E_GTD, E_NSLOT };
// the work space of this kernel, wavefront-major: level k of the 64 columns of wavefront b holds its E_NSLOT values
// and the interface pressure in WNS consecutive rows of 64 doubles, at wk + (((b*(kk+1) + k-1)*WNS + slot)*64 + lane).
// A sweep reads and writes 3-8 slots of a level: with the slots of a level side by side they are one 6 KB piece of
// memory instead of pieces of 512 B that lie 45 MB apart.
#define WNS (E_NSLOT + 1)
#define W(slot, k) wb[((size_t)((k)-1) * WNS + (slot)) * 64]
#define ST(a, k) (a)[c + (size_t)((k)-1) * np]
#define SIGR(k) sigr[c + (size_t)((k)-1) * np]
#define PRES(k) wb[((size_t)((k)-1) * WNS + E_NSLOT) * 64]
#define TRC(nt, k) trc[c + (size_t)((k)-1 + (nt)*2 * kk) * np]

// one wavefront per 64 columns and ~1700 wavefronts in all: occupancy cannot exceed 2 waves per SIMD anyway,
// so let the register allocator use up to 256 VGPRs instead of spilling at 128
// DU: levels whose loads the light sweeps keep in flight; the two fused sweeps (stratification + limiter down, limiter
// up + first guess) carry an equation-of-state evaluation resp. a square root and four divisions per level and are
// unrolled DH = min(DU, 4) deep: 8 deep they need 255 VGPRs and spill 300 SGPRs
template <int DU>
__global__ void __launch_bounds__(64, 1) k_diapfl_column3(const DevView *__restrict__ Vp, int n, int nn, int *__restrict__ errflag KPROF_ARGS) {
  const DevView &V = *Vp;
  constexpr int DH = DU < 4 ? DU : 4;
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  const size_t c = t_;
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk, ntr = V.ntr;
  const Params P = V.P;        // by value: the equation-of-state coefficients stay in registers across the stores
  const double dsgmnr = .1, fcmxr = .25, dsgcr0 = .25, dfeps = 1.e-12, gbbl = .2, kappa = .4, ustmin = .0001;
  const double cc = GRAV * GRAV * P.delt1 / (ALPHA0 * ALPHA0);                       // :95
  [[maybe_unused]] const int wid = blockIdx.x;     // KPROF words: 0 start, 1 limiter starts, 2 solver starts, 3 mixing starts, 4 massless layers start, 5 end; 6 limiter sweeps, 7 solver iterations (summed over lanes)
  KPROF_MARK(wid, 0);
  // (address-space typed: blomgpu_internal.h, PtrTable)
  gcd_t __restrict__ sigr = V.f[F_sigmar];
  gd_t __restrict__ wb = global_ptr(V.wk) + (size_t)blockIdx.x * (kk + 1) * WNS * 64 + threadIdx.x;
  gd_t __restrict__ temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
  gd_t __restrict__ dp = V.f[F_dp] + (size_t)nn * np, sigma = V.f[F_sigma] + (size_t)nn * np;
  gd_t __restrict__ trc = V.f[F_trc] + (size_t)nn * np, nu = V.f[F_difdia];
  gd_t __restrict__ fpug = V.f[F_fpug], fplg = V.f[F_fplg];

  int kmax = 1;                                                                      // :139-143
  for (int k0 = 2; k0 <= kk; k0 += 2 * DU) {
    double a0[2 * DU];
#pragma unroll
    for (int u = 0; u < 2 * DU; u++) a0[u] = ST(dp, CLAMPK(k0 + u, 2, kk));
#pragma unroll
    for (int u = 0; u < 2 * DU; u++)
      if (k0 + u <= kk && a0[u] > EPSILP) kmax = k0 + u;
  }
  const int kfpl = V.m[I_kfpla][c + (size_t)(n - 1) * np];
  const int kmin = kfpl - 2;
  double fpl0 = 0., fpl1 = 0.;      // fpl(kmin) = fpu(kmin+1), fpl(kmin+1) [= fpu(kfpl)]
  const bool mixing = kmin < kmax;
  if (mixing) {
    const double d1 = ST(dp, 1), d2 = ST(dp, 2);
    // interface pressures of the compacted column, :175-178 (positions kmin, kmin+1 are layers 1, 2)
    double acc = 0.;
    acc = acc + d1;
    const double p1 = acc;                                // pres(kmin+1)
    acc = acc + d2;
    const double preskf = acc;                            // pres(kmin+2) = pres(kfpl)
    PRES(kfpl) = acc;
    for (int k0 = kfpl; k0 <= kmax; k0 += 2 * DU) {
      double a0[2 * DU];
#pragma unroll
      for (int u = 0; u < 2 * DU; u++) a0[u] = ST(dp, CLAMPK(k0 + u, kfpl, kmax));
#pragma unroll
      for (int u = 0; u < 2 * DU; u++)
        if (k0 + u <= kmax) { acc = acc + a0[u]; PRES(k0 + u + 1) = acc; }
    }
    const double presb = acc;
    {                                                                                // :182-193
      fpl0 = fmin2(fmin2(p1, presb - p1), cc * ST(nu, 1) * (d1 + d2) / (2. * d1 * d2));
      const double delpu = fmax2(ONEM, d2), delpl = fmax2(ONEM, ST(dp, kfpl <= kk ? kfpl : kk));
      fpl1 = fmin2(fmin2(preskf, presb - preskf), cc * ST(nu, 2) * (delpu + delpl) / (2. * delpu * delpl));
      if (kmax == kmin + 1) fpl1 = 0.;                    // fpl(kmax) = 0 comes last in the reference
    }
    if (kfpl <= kmax) {
      int k;
      if (kfpl < kmax) {                                                             // :197-209
        k = kmax - 1;
        const double us = V.f[F_ustarb][c];
        const double nubbl = gbbl * (us * us * us) *
                             exp_libm(-(ST(dp, k + 1) + .5 * ST(dp, k)) * fabs(V.f[F_coriop][c]) * ALPHA0 /
                                 (kappa * fmax2(ustmin, us) * GRAV)) /
                             (ALPHA0 * GRAV * (SIGR(k + 1) - SIGR(k)));
        ST(nu, k) = fmax2(ST(nu, k), nubbl);
      }
      // ---- stratification and density-restoring fluxes, :217-287, and the flux limiter's first downward sweep, :292-305,
      //      in ONE pass from the bottom up: the stratification of a level is a stencil over the levels k-1, k, k+1 (no
      //      recurrence), so it can be evaluated in the order the limiter's downward sweep visits the levels, and the
      //      limiter takes dsgu, dsgl, fcu, fcl of the level from registers instead of reading them back.
      const int rst1 = kfpl;
      int rst2 = -1;
      if (kfpl != kmax)
        if (ST(sigma, kfpl) > .5 * (SIGR(kfpl) + SIGR(kfpl + 1))) rst2 = kfpl + 1;
      // position kfpl-1 holds layer 2: its T, S are layer 2's; its density and reference density stay those of kfpl-1
#define TIX(kq) ((kq) == kfpl - 1 ? 2 : (kq))
      double tp = ST(temp, kmax), sp = ST(saln, kmax), dens_p = ST(sigma, kmax), sr_p = SIGR(kmax);               // level k+1
      double tk = ST(temp, TIX(kmax - 1)), sk = ST(saln, TIX(kmax - 1)), dens_k = ST(sigma, kmax - 1), sr_k = SIGR(kmax - 1);
      // k = kmax, :275-287 (its own values are the sweep's first "level k+1", position kmax-1 its first "level k")
      double su_b, sui_b, fpu_b, fcu_b;
      {
        const double dsgdt = eosd::dsigdt(P, tp, sp), dsgds = eosd::dsigds(P, tp, sp);
        su_b = fmax2(dsgmnr * (sr_p - sr_k), dsgdt * (tp - tk) + dsgds * (sp - sk));
        sui_b = 1. / su_b;
        const double dkm1 = kmax - 1 == kmin + 1 ? d2 : ST(dp, kmax - 1);
        if (dens_p > sr_p && dens_k < sr_p) fpu_b = fmin2(dkm1, (dens_p - sr_p) * ST(dp, kmax) * sui_b);
        else fpu_b = 0.;
        fcu_b = fpu_b * su_b;
      }
      // ---- flux limiter, :292-330 ------------------------------------------------------------
      KPROF_MARK(wid, 1);
      bool done = false, first = true;
      int niter = 0, kfmaxu = 0;
      double dflim = 0.;
      while (!done) {
        done = true;
        double fmax_p = 0., fcu_p = fcu_b, sui_p = sui_b;
        if (first) {
          first = false;
          for (int k0 = kmax - 1; k0 >= kfpl; k0 -= DH) {
            double a0[DH], a1[DH], a2[DH], a3[DH], a4[DH], a5[DH], a6[DH];
#pragma unroll
            for (int u = 0; u < DH; u++) {
              const int kq = CLAMPK(k0 - u, kfpl, kmax - 1);
              a0[u] = ST(temp, TIX(kq - 1)); a1[u] = ST(saln, TIX(kq - 1)); a2[u] = ST(sigma, kq - 1); a3[u] = SIGR(kq - 1);
              a4[u] = ST(dp, kq); a5[u] = ST(nu, kq); a6[u] = PRES(kq + 1);
            }
#pragma unroll
            for (int u = 0; u < DH; u++) {
              k = k0 - u;
              if (k < kfpl) break;
              const double tm = a0[u], sm = a1[u], dens_m = a2[u], sr_m = a3[u];
              double su = 1., sl = 1., fcu = 0., fcl = 0.;
              if (k != rst1 && k != rst2) {
                const double dk = a4[u], nuk = a5[u];
                const double dsgdt = eosd::dsigdt(P, tk, sk), dsgds = eosd::dsigds(P, tk, sk);
                su = fmax2(dsgmnr * (sr_k - sr_m), dsgdt * (tk - tm) + dsgds * (sk - sm));
                sl = fmax2(dsgmnr * (sr_p - sr_k), dsgdt * (tp - tk) + dsgds * (sp - sk));
                const double shm = 2. * su * sl / (su + sl);
                const double sg = .5 * (su + sl);
                const double sui = 1. / su, sli = 1. / sl;
                const double fcmx = .25 * (sqrt(dk * dk + 4. * cc * nuk * sg * (sui + sli)) - dk) * shm * fcmxr;
                const double dsgc = dens_k - sr_k;
                if (dsgc > 0.) {
                  if (dens_m < sr_k) {
                    double q = fmax2(0., (dens_k - sr_p) / ((sr_k - sr_p) * (1. - dsgcr0)));
                    q = fmax2(0., 1. - q * q);
                    q = q * q * q;
                    fcu = dsgc * dk;
                    fcu = fmin2(q * fcu + (1. - q) * fcmx, fcu);
                  }
                } else {
                  if (dens_p > sr_k) {
                    double q = fmax2(0., (dens_k - sr_m) / ((sr_k - sr_m) * (1. - dsgcr0)));
                    q = fmax2(0., 1. - q * q);
                    q = q * q * q;
                    fcl = dsgc * dk;
                    fcl = fmax2(q * fcl - (1. - q) * fcmx, fcl);
                  }
                }
              }
              // the limiter's step for this level, :296-304
              const double q = ((fmax_p + fcu_p) * sui_p + presb - a6[u]) * sl;
              fcl = fmax2(-q, fcl);
              fmax_p = q + fcl;
              W(E_SU, k) = su; W(E_SL, k) = sl; W(E_FCU, k) = fcu; W(E_FCL, k) = fcl; W(E_FMAX, k) = fmax_p;
              fcu_p = fcu;
              sui_p = 1. / su;
              tp = tk; sp = sk; dens_p = dens_k; sr_p = sr_k;
              tk = tm; sk = sm; dens_k = dens_m; sr_k = sr_m;
            }
          }
        } else
        for (int k0 = kmax - 1; k0 >= kfpl; k0 -= DU) {
          double a0[DU], a1[DU], a2[DU], a3[DU], a4[DU];
#pragma unroll
          for (int u = 0; u < DU; u++) {
            const int kq = CLAMPK(k0 - u, kfpl, kmax - 1);
            a0[u] = PRES(kq + 1); a1[u] = W(E_SL, kq); a2[u] = W(E_FCL, kq); a3[u] = W(E_FCU, kq); a4[u] = W(E_SU, kq);
          }
#pragma unroll
          for (int u = 0; u < DU; u++) {
            k = k0 - u;
            if (k >= kfpl) {
              const double q = ((fmax_p + fcu_p) * sui_p + presb - a0[u]) * a1[u];
              const double fcl = fmax2(-q, a2[u]);
              W(E_FCL, k) = fcl;
              fmax_p = q + fcl;
              W(E_FMAX, k) = fmax_p;
              fcu_p = a3[u];
              sui_p = 1. / a4[u];
            }
          }
        }
        kfmaxu = 0;
        // The upward sweep, :306-316, carries the first guess, :334-353, one level behind it: level k-1's guess needs
        // fcu and dsgu of level k, final once the sweep has visited it, and otherwise what the sweep holds of level k-1
        // in registers.  A sweep that has to be repeated (done = false) leaves guesses that the next one overwrites.
        double fmax_m = 0., fcl_m = -fpl1, sli_m = 1.;
        double g_fcl_m = -fpl1, g_sli_m = 1.;                     // the first guess's fcl, 1/dsgl of level k-2
        double v_su = 1., v_sl = 1., v_fcu = 0., v_fm = 0., v_fcl = 0., v_nu = 0., v_dp = 0.;   // level k-1
        dflim = 0.;
#define DIAPFL_GUESS(kl, fcu_n, su_n)                                                                                     \
        {                                                                                                                 \
          const double su = v_su, sl = v_sl;                                                                              \
          const double shm = 2. * su * sl / (su + sl);                                                                    \
          const double sg = .5 * (su + sl);                                                                               \
          const double sui = 1. / su, sli = 1. / sl;                                                                      \
          const double fk = fmin2(fmin2(v_fm, .5 * sqrt(cc * v_nu * sg * (sui + sli)) * shm), cc * v_nu * sg / fmax2(EPSILP, v_dp)); \
          W(E_F, kl) = fk;                                                                                                \
          W(E_H, kl) = v_fcu * sui - v_fcl * sli + g_fcl_m * g_sli_m - (fcu_n) * (1. / (su_n));                          \
          W(E_R, kl) = 4. * cc * v_nu * sg * (sui + sli);                                                                 \
          W(E_T, kl) = .25 * shm;                                                                                         \
          dflim = fmax2(dflim, v_fm);                                                                                     \
          g_fcl_m = v_fcl; g_sli_m = sli;                                                                                 \
        }
        for (int k0 = kfpl; k0 <= kmax - 1; k0 += DH) {
          double a0[DH], a1[DH], a2[DH], a3[DH], a4[DH], a5[DH], a6[DH], a7[DH];
#pragma unroll
          for (int u = 0; u < DH; u++) {
            const int kq = CLAMPK(k0 + u, kfpl, kmax - 1);
            a0[u] = PRES(kq); a1[u] = W(E_SU, kq); a2[u] = W(E_FCU, kq); a3[u] = W(E_FMAX, kq); a4[u] = W(E_FCL, kq); a5[u] = W(E_SL, kq);
            a6[u] = ST(nu, kq); a7[u] = ST(dp, kq);
          }
#pragma unroll
          for (int u = 0; u < DH; u++) {
            k = k0 + u;
            if (k <= kmax - 1) {
              const double q = ((fmax_m - fcl_m) * sli_m + a0[u] - preskf) * a1[u];
              double fcu = a2[u];
              if (fcu > q) { fcu = q; W(E_FCU, k) = q; done = false; }
              double fm = a3[u];
              if (fm > q - fcu) { fm = q - fcu; W(E_FMAX, k) = fm; kfmaxu = k; }
              fmax_m = fm;
              fcl_m = a4[u];
              sli_m = 1. / a5[u];
              if (k > kfpl) DIAPFL_GUESS(k - 1, fcu, a1[u])
              v_su = a1[u]; v_sl = a5[u]; v_fcu = fcu; v_fm = fm; v_fcl = a4[u]; v_nu = a6[u]; v_dp = a7[u];
            }
          }
        }
        if (kfpl <= kmax - 1) DIAPFL_GUESS(kmax - 1, fcu_b, su_b)
#undef DIAPFL_GUESS
        // the reference tests niter == 100 without ever incrementing niter in this loop (:317),
        // i.e. it never aborts here; we bound the loop defensively and flag it.
        KPROF_ADD(wid, 6, 1);
        if (++niter > 100000) { atomicOr(errflag, 1); break; }
      }
      dflim = dflim * dfeps;
      // ---- implicit solve by alternating sweeps, :357-533 ------------------------------------
      KPROF_MARK(wid, 2);
      niter = 0;
      bool dwnwrd = false;
      for (;;) {
        dwnwrd = !dwnwrd;
        double maxdf = 0., ctd = 0., atd = 0., bitd = 1.;
        bool remfmx = false;
        if (dwnwrd) {
          double f0m = 0., fnew_m = 0., slim = 1.;             // f0, f, dsgli of level kfpl-1
          for (int k0 = kfpl; k0 <= kmax - 1; k0 += DU) {
            double a0[DU], a1[DU], a2[DU], a3[DU], a4[DU], a5[DU], a6[DU], a7[DU];
#pragma unroll
            for (int u = 0; u < DU; u++) {
              const int kq = CLAMPK(k0 + u, kfpl, kmax - 1), kn = CLAMPK(kq + 1, kfpl, kmax - 1);
              a0[u] = W(E_FMAX, kq); a1[u] = W(E_F, kn); a2[u] = W(E_SU, kn); a3[u] = ST(dp, kq); a4[u] = W(E_H, kq);
              a5[u] = W(E_R, kq); a6[u] = W(E_T, kq); a7[u] = W(E_SL, kq);
            }
#pragma unroll
            for (int u = 0; u < DU; u++) {
              k = k0 + u;
              if (k <= kmax - 1) {
                const double fmx = a0[u];
                if (remfmx) { W(E_GTD, k) = 0.; W(E_FT, k) = fmx; }
                else {
                  const double fp = k + 1 <= kmax - 1 ? a1[u] : 0.;
                  const double suip = k + 1 <= kmax - 1 ? 1. / a2[u] : sui_b;
                  const double q = f0m * slim + fp * suip - a3[u] - a4[u];
                  double f0, dfdg;
                  flux_solution(q, a5[u], a6[u], f0, dfdg);
                  if (f0 >= fmx) { f0 = fmx; dfdg = 0.; if (k > kfmaxu) remfmx = true; }
                  const double g = ctd * bitd;
                  W(E_GTD, k) = g;
                  atd = -dfdg * slim;
                  ctd = -dfdg * suip;
                  bitd = 1. / (1. - atd * g);
                  fnew_m = (f0 - atd * (fnew_m - f0m) + ctd * fp) * bitd;
                  W(E_FT, k) = fnew_m;
                  f0m = f0;
                  slim = 1. / a7[u];
                }
              }
            }
          }
          double fnew_p = 0., gtd_p = 0.;                       // f, gtd of level kmax
          for (int k0 = kmax - 1; k0 >= kfpl; k0 -= DU) {
            double a0[DU], a1[DU], a2[DU], a3[DU];
#pragma unroll
            for (int u = 0; u < DU; u++) {
              const int kq = CLAMPK(k0 - u, kfpl, kmax - 1);
              a0[u] = W(E_FMAX, kq); a1[u] = W(E_FT, kq); a2[u] = W(E_F, kq); a3[u] = W(E_GTD, kq);
            }
#pragma unroll
            for (int u = 0; u < DU; u++) {
              k = k0 - u;
              if (k >= kfpl) {
                const double fk = fmin2(a0[u], a1[u] - gtd_p * fnew_p);
                maxdf = fmax2(maxdf, fabs(fk - a2[u]));
                W(E_F, k) = fk;
                fnew_p = fk;
                gtd_p = a3[u];
              }
            }
          }
        } else {
          double f0p = 0., fnew_p = 0., suip = sui_b;           // f0, f of level kmax; dsgui(kmax)
          for (int k0 = kmax - 1; k0 >= kfpl; k0 -= DU) {
            double a0[DU], a1[DU], a2[DU], a3[DU], a4[DU], a5[DU], a6[DU], a7[DU];
#pragma unroll
            for (int u = 0; u < DU; u++) {
              const int kq = CLAMPK(k0 - u, kfpl, kmax - 1), km1 = CLAMPK(kq - 1, kfpl, kmax - 1);
              a0[u] = W(E_FMAX, kq); a1[u] = W(E_F, km1); a2[u] = W(E_SL, km1); a3[u] = ST(dp, kq); a4[u] = W(E_H, kq);
              a5[u] = W(E_R, kq); a6[u] = W(E_T, kq); a7[u] = W(E_SU, kq);
            }
#pragma unroll
            for (int u = 0; u < DU; u++) {
              k = k0 - u;
              if (k >= kfpl) {
                const double fmx = a0[u];
                if (remfmx) { W(E_GTD, k) = 0.; W(E_FT, k) = fmx; }
                else {
                  const double fm = k - 1 >= kfpl ? a1[u] : 0.;
                  const double slim = k - 1 >= kfpl ? 1. / a2[u] : 1.;
                  const double q = fm * slim + f0p * suip - a3[u] - a4[u];
                  double f0, dfdg;
                  flux_solution(q, a5[u], a6[u], f0, dfdg);
                  if (f0 >= fmx) { f0 = fmx; dfdg = 0.; if (k <= kfmaxu) remfmx = true; }
                  const double g = atd * bitd;
                  W(E_GTD, k) = g;
                  atd = -dfdg * slim;
                  ctd = -dfdg * suip;
                  bitd = 1. / (1. - ctd * g);
                  fnew_p = (f0 + atd * fm - ctd * (fnew_p - f0p)) * bitd;
                  W(E_FT, k) = fnew_p;
                  f0p = f0;
                  suip = 1. / a7[u];
                }
              }
            }
          }
          double fnew_m = 0., gtd_m = 0.;                       // f, gtd of level kfpl-1
          for (int k0 = kfpl; k0 <= kmax - 1; k0 += DU) {
            double a0[DU], a1[DU], a2[DU], a3[DU];
#pragma unroll
            for (int u = 0; u < DU; u++) {
              const int kq = CLAMPK(k0 + u, kfpl, kmax - 1);
              a0[u] = W(E_FMAX, kq); a1[u] = W(E_FT, kq); a2[u] = W(E_F, kq); a3[u] = W(E_GTD, kq);
            }
#pragma unroll
            for (int u = 0; u < DU; u++) {
              k = k0 + u;
              if (k <= kmax - 1) {
                const double fk = fmin2(a0[u], a1[u] - gtd_m * fnew_m);
                maxdf = fmax2(maxdf, fabs(fk - a2[u]));
                W(E_F, k) = fk;
                fnew_m = fk;
                gtd_m = a3[u];
              }
            }
          }
        }
        niter = niter + 1;
        KPROF_ADD(wid, 7, 1);
        if (maxdf <= dflim) break;
        if (niter == 100) { atomicOr(errflag, 2); break; }                         // :520-532 (xchalt)
      }
      // ---- interface fluxes, :536-541 -------------------------------------------------------------
      for (int k0 = kfpl; k0 <= kmax - 1; k0 += DU) {
        double a0[DU], a1[DU], a2[DU], a3[DU], a4[DU];
#pragma unroll
        for (int u = 0; u < DU; u++) {
          const int kq = CLAMPK(k0 + u, kfpl, kmax - 1);
          a0[u] = W(E_F, kq); a1[u] = W(E_FCU, kq); a2[u] = W(E_SU, kq); a3[u] = W(E_FCL, kq); a4[u] = W(E_SL, kq);
        }
#pragma unroll
        for (int u = 0; u < DU; u++) {
          k = k0 + u;
          if (k <= kmax - 1) {
            const double fk = a0[u];
            ST(fpug, k) = (fk + a1[u]) * (1. / a2[u]);
            ST(fplg, k) = (fk - a3[u]) * (1. / a4[u]);
          }
        }
      }
      ST(fpug, kmax) = fpu_b;
      ST(fplg, kmax) = 0.;
      ST(fpug, kfpl) = fpl1;                                                         // :541
    }
    KPROF_MARK(wid, 3);
    // ---- implicit mixing of S, T, tracers over positions kmin..kmax, :546-576, fused with the layer
    //      thickness update :572-576 (which only reads fluxes) ------------------------------------
    {
      const bool interior = kfpl <= kmax;
      const bool more_tracers = ntr > MAXTR;               // tracers beyond the first MAXTR: same system, coefficients from the work space
      double ctd = 0., bitd = 1., g = 0.;
      const int km1 = kmin - 1 > 1 ? kmin - 1 : 1;
      double s_prev = ST(saln, km1), t_prev = ST(temp, km1);
      double tr_prev[MAXTR];
#pragma unroll
      for (int nt = 0; nt < MAXTR; nt++) tr_prev[nt] = nt < ntr ? TRC(nt, km1) : 0.;
      double fl_m = 0.;                                   // fpl of the previous position
      double fu_next = interior ? ST(fpug, kfpl) : 0.;
      for (int p0 = kmin; p0 <= kmax; p0 += DU) {
       double b0[DU], b1[DU], b2[DU], b3[DU], b4[DU], bt[MAXTR][DU];
#pragma unroll
       for (int u = 0; u < DU; u++) {
         const int pq = CLAMPK(p0 + u, kmin, kmax);
         const int lq = pq == kmin ? 1 : (pq == kmin + 1 ? 2 : pq);
         b0[u] = ST(dp, lq); b1[u] = ST(fplg, lq); b2[u] = ST(fpug, CLAMPK(pq + 1, 1, kk)); b3[u] = ST(saln, lq); b4[u] = ST(temp, lq);
#pragma unroll
         for (int nt = 0; nt < MAXTR; nt++) bt[nt][u] = nt < ntr ? TRC(nt, lq) : 0.;
       }
#pragma unroll
       for (int u = 0; u < DU; u++) {
        const int pos = p0 + u;
        if (pos > kmax) break;
        const int lay = pos == kmin ? 1 : (pos == kmin + 1 ? 2 : pos);
        double dk, fu, fl;
        if (pos == kmin) { dk = d1; fu = 0.; fl = fpl0; }
        else if (pos == kmin + 1) { dk = d2; fu = fpl0; fl = fpl1; }
        else { dk = b0[u]; fu = fu_next; fl = b1[u]; }
        g = ctd * bitd;
        W(E_GTD, pos) = g;
        const double q = 1. / (dk + fu + fl);
        const double atd = -fu * q;
        ctd = -fl * q;
        const double dtd = dk * q;
        bitd = 1. / (1. - atd * g);
        if (more_tracers) { W(E_R, pos) = dtd; W(E_T, pos) = atd; W(E_F, pos) = bitd; }   // the solve's slots are free by now
        s_prev = (dtd * b3[u] - atd * s_prev) * bitd;
        t_prev = (dtd * b4[u] - atd * t_prev) * bitd;
        ST(saln, lay) = s_prev;
        ST(temp, lay) = t_prev;
#pragma unroll
        for (int nt = 0; nt < MAXTR; nt++)
          if (nt < ntr) { tr_prev[nt] = (dtd * bt[nt][u] - atd * tr_prev[nt]) * bitd; TRC(nt, lay) = tr_prev[nt]; }
        if (pos >= kfpl) {                                // interior layers: new thickness
          if (pos < kmax) {
            fu_next = b2[u];
            ST(dp, pos) = fmax2(0., dk + fu + fl - fl_m - fu_next);
          } else
            ST(dp, pos) = fmax2(0., dk + fu - fl_m);
        } else if (pos == kmax && kmin <= 2)              // kmax = kmin+1: layer 2 keeps the update only
          ST(dp, 2) = fmax2(0., dk + fu - fl_m);          // when its position is copied back (:593-599)
        fl_m = fl;
       }
      }
      double s_next = s_prev, t_next = t_prev, g_next = g;
      double tr_next[MAXTR];
#pragma unroll
      for (int nt = 0; nt < MAXTR; nt++) tr_next[nt] = tr_prev[nt];
      for (int p0 = kmax - 1; p0 >= kmin; p0 -= DU) {
       double b0[DU], b3[DU], b4[DU], bt[MAXTR][DU];
#pragma unroll
       for (int u = 0; u < DU; u++) {
         const int pq = CLAMPK(p0 - u, kmin, kmax - 1);
         const int lq = pq == kmin ? 1 : (pq == kmin + 1 ? 2 : pq);
         b0[u] = W(E_GTD, pq); b3[u] = ST(saln, lq); b4[u] = ST(temp, lq);
#pragma unroll
         for (int nt = 0; nt < MAXTR; nt++) bt[nt][u] = nt < ntr ? TRC(nt, lq) : 0.;
       }
#pragma unroll
       for (int u = 0; u < DU; u++) {
        const int pos = p0 - u;
        if (pos < kmin) break;
        const int lay = pos == kmin ? 1 : (pos == kmin + 1 ? 2 : pos);
        s_next = b3[u] - g_next * s_next;
        t_next = b4[u] - g_next * t_next;
        ST(saln, lay) = s_next;
        ST(temp, lay) = t_next;
        ST(sigma, lay) = eos::sig(P, t_next, s_next);
#pragma unroll
        for (int nt = 0; nt < MAXTR; nt++)
          if (nt < ntr) { tr_next[nt] = bt[nt][u] - g_next * tr_next[nt]; TRC(nt, lay) = tr_next[nt]; }
        g_next = b0[u];
       }
      }
      // the tracers beyond the first MAXTR, MAXTR at a time: the same two sweeps with the stored coefficients
      for (int nt0 = MAXTR; nt0 < ntr; nt0 += MAXTR) {
        double xp[MAXTR];
#pragma unroll
        for (int b = 0; b < MAXTR; b++) xp[b] = nt0 + b < ntr ? TRC(nt0 + b, km1) : 0.;
        for (int pos = kmin; pos <= kmax; pos++) {
          const int lay = pos == kmin ? 1 : (pos == kmin + 1 ? 2 : pos);
          const double dtd = W(E_R, pos), atd = W(E_T, pos), bi = W(E_F, pos);
          double xv[MAXTR];
#pragma unroll
          for (int b = 0; b < MAXTR; b++) xv[b] = nt0 + b < ntr ? TRC(nt0 + b, lay) : 0.;
#pragma unroll
          for (int b = 0; b < MAXTR; b++)
            if (nt0 + b < ntr) { xp[b] = (dtd * xv[b] - atd * xp[b]) * bi; TRC(nt0 + b, lay) = xp[b]; }
        }
        double gn = g;
        for (int pos = kmax - 1; pos >= kmin; pos--) {
          const int lay = pos == kmin ? 1 : (pos == kmin + 1 ? 2 : pos);
          const double gp = W(E_GTD, pos);
          double xv[MAXTR];
#pragma unroll
          for (int b = 0; b < MAXTR; b++) xv[b] = nt0 + b < ntr ? TRC(nt0 + b, lay) : 0.;
#pragma unroll
          for (int b = 0; b < MAXTR; b++)
            if (nt0 + b < ntr) { xp[b] = xv[b] - gn * xp[b]; TRC(nt0 + b, lay) = xp[b]; }
          gn = gp;
        }
      }
    }
    // dens is the one work array the reference does not move with the mixed layer (:159-172): when the
    // column has no interior mass (kmax = kmin+1) position kmin+1 is not re-evaluated by the sweep
    // above and the copy-back :580-599 hands layer 2 the old density of layer kmin+1
    if (kmax == kmin + 1 && kmin >= 2) ST(sigma, 2) = ST(sigma, kmin + 1);
  }
  // ---- massless layers, :605-651 ---------------------------------------------------------------
  KPROF_MARK(wid, 4);
  if (kfpl > kmax) {
    const double t2 = ST(temp, 2);
    for (int k = 3; k <= kk; k++) {
      const double tk = fmax2(t2, V.f[F_temmin][c + (size_t)(k - 1) * np]);
      ST(temp, k) = tk;
      ST(sigma, k) = SIGR(k);
      ST(saln, k) = eosd::sofsig(P, SIGR(k), tk);
      ST(dp, k) = 0.;
      for (int nt = 0; nt < ntr; nt++) TRC(nt, k) = TRC(nt, 2);
    }
  } else {
    const double tf = ST(temp, kfpl);
    double trf[MAXTR], trm[MAXTR];
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++) { trf[nt] = nt < ntr ? TRC(nt, kfpl) : 0.; trm[nt] = nt < ntr ? TRC(nt, kmax) : 0.; }
    const double tm = ST(temp, kmax);
    for (int k0 = 3; k0 <= kfpl - 1; k0 += 2 * DU) {
      double a0[2 * DU];
#pragma unroll
      for (int u = 0; u < 2 * DU; u++) a0[u] = SIGR(CLAMPK(k0 + u, 3, kfpl - 1));
#pragma unroll
      for (int u = 0; u < 2 * DU; u++) {
        const int k = k0 + u;
        if (k > kfpl - 1) break;
        ST(temp, k) = tf;
        ST(sigma, k) = a0[u];
        ST(saln, k) = eosd::sofsig(P, a0[u], tf);
        ST(dp, k) = 0.;
#pragma unroll
        for (int nt = 0; nt < MAXTR; nt++)
          if (nt < ntr) TRC(nt, k) = trf[nt];
      }
    }
    for (int nt = MAXTR; nt < ntr; nt++) {
      const double xf = TRC(nt, kfpl), xm = TRC(nt, kmax);
      for (int k = 3; k <= kfpl - 1; k++) TRC(nt, k) = xf;
      for (int k = kmax + 1; k <= kk; k++) TRC(nt, k) = xm;
    }
    for (int k0 = kmax + 1; k0 <= kk; k0 += 2 * DU) {
      double a0[2 * DU];
#pragma unroll
      for (int u = 0; u < 2 * DU; u++) a0[u] = SIGR(CLAMPK(k0 + u, kmax + 1, kk));
#pragma unroll
      for (int u = 0; u < 2 * DU; u++) {
        const int k = k0 + u;
        if (k > kk) break;
        ST(temp, k) = tm;
        ST(sigma, k) = a0[u];
        ST(saln, k) = eosd::sofsig(P, a0[u], tm);
#pragma unroll
        for (int nt = 0; nt < MAXTR; nt++)
          if (nt < ntr) TRC(nt, k) = trm[nt];
      }
    }
  }
  // ---- lower bounds of the TKE / generic-length-scale tracers (:612-626 in the massless layers, :662-677 on
  //      every layer at the copy-back; max is idempotent, so one pass over the column covers both) ----------
  if (P.itrtke >= 1) {
    const int a = P.itrtke - 1, b = P.itrgls - 1;
    const bool doa = a < ntr, dob = P.gls && b >= 0 && b < ntr;
    for (int k0 = 1; k0 <= kk; k0 += 2 * DU) {
      double a0[2 * DU], a1[2 * DU];
#pragma unroll
      for (int u = 0; u < 2 * DU; u++) {
        const int kq = CLAMPK(k0 + u, 1, kk);
        a0[u] = doa ? TRC(a, kq) : 0.;
        a1[u] = dob ? TRC(b, kq) : 0.;
      }
#pragma unroll
      for (int u = 0; u < 2 * DU; u++) {
        const int k = k0 + u;
        if (k > kk) break;
        if (doa) TRC(a, k) = fmax2(a0[u], TKE_MIN);
        if (dob) TRC(b, k) = fmax2(a1[u], GLS_PSI_MIN);
      }
    }
  }
  // ---- interface pressure and the fluxes handed to the momentum mixing, :654-700, :718 ---------
  column_scan(V.f[F_p][c], dp + c, V.f[F_p] + c, np, kk);
  V.f[F_util1][c] = (double)kmin;
  if (mixing) {
    for (int k = 1; k <= kmin; k++) { ST(fpug, k) = fpl0; ST(fplg, k) = fpl0; }
    ST(fpug, kmin + 1) = fpl0;
    ST(fplg, kmin + 1) = fpl1;
    for (int k = kmax + 1; k <= kk; k++) { ST(fpug, k) = 0.; ST(fplg, k) = 0.; }
  } else
    for (int k = 1; k <= kk; k++) { ST(fpug, k) = 0.; ST(fplg, k) = 0.; }
  KPROF_MARK(wid, 5);
}

int diapfl_column3_launch(blomgpu_ctx *c, int n, int nn, int *errflag) {
  const DevView &h = c->h;
  if (E_NSLOT > h.nwk) return ctx_fail(c, "diapfl: work space too small");
  const dim3 g = plane_grid(h, 1, 64);
  if ((size_t)g.x * (h.kk + 1) * WNS * 64 > (size_t)h.nwk * h.kk * h.nplane) return ctx_fail(c, "diapfl: work space too small");
  TimeScope tk(c, "k_diapfl_column3");
  if (c->diapfl_du == 8) hipLaunchKernelGGL(k_diapfl_column3<8>, g, dim3(64), 0, c->stream, c->d, n, nn, errflag KPROF_PASS(2));
  else if (c->diapfl_du == 2) hipLaunchKernelGGL(k_diapfl_column3<2>, g, dim3(64), 0, c->stream, c->d, n, nn, errflag KPROF_PASS(2));
  else hipLaunchKernelGGL(k_diapfl_column3<4>, g, dim3(64), 0, c->stream, c->d, n, nn, errflag KPROF_PASS(2));
  return 0;
}
