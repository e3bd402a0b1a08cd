// diapfl -- diapycnal mixing (isopycnic bulk-mixed-layer branch), phy/mod_diapfl.F90:49-1046.
//
// A per-water-column algorithm: mixed-layer/interior mass fluxes, density-restoring flux
// corrections with an iterative limiter (:296-330), an implicit nonlinear diffusion solve by
// alternating downward/upward tridiagonal sweeps (:359-533), implicit T/S/tracer mixing
// (:546-576), then per u-/v-column implicit momentum mixing (:740-966) and dpu/dpv (:971-1000).
// Mapping: one thread per column, columns of a wavefront adjacent in i.  The reference's
// ~22 private 1-D work arrays of length kdm become kk-level work-space planes, so every
// "array(k)" access of a wavefront is one coalesced plane-row access instead of per-lane
// scratch memory.  The trip counts of the two iteration loops are data dependent; lanes that
// converged early idle until their wavefront neighbours finish (hard part 7).
// Algorithmic bytes: (23 + 2*ntr) F; roofline: HBM (this first version adds the work-plane traffic).
#include "blomgpu_internal.h"
#include "eos.h"

#define GRAV 9.806
#define ALPHA0 1.e-3
#define EPSILP 1.e-12
#define ONEM 9806.
#define MAXTR 8

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

enum { D_TTEM, D_SSAL, D_DELP, D_DENS, D_NU, D_FPU, D_FPL, D_FCU, D_FCL, D_DSGU, D_DSGL, D_DSGHM, D_DSG,
       D_DSGUI, D_DSGLI, D_FMAX, D_F, D_F0, D_FOLD, D_H, D_GTD, D_TTRC };
// 1-based layer index k, as in the Fortran
#define AR(slot, k) WK(V, slot)[c + (size_t)((k)-1) * np]

#include "diapfl_common.h"

__global__ void k_diapfl_column(const DevView *__restrict__ Vp, int n, int nn, int *__restrict__ errflag) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int kk = V.kk, ntr = V.ntr;
  const Params &P = V.P;
  const double dsgmnr = .1, fcmxr = .25, dsgcr0 = .25, dfeps = 1.e-12, gbbl = .2, kappa = .4, ustmin = .0001;
  const double cc = GRAV * GRAV * P.delt1 / (ALPHA0 * ALPHA0);                       // :95
  const double *sigr = V.f[F_sigmar];
#define SIGR(k) sigr[c + (size_t)((k)-1) * np]
  double *pres = V.f[F_wkp1];
#define PRES(k) pres[c + (size_t)((k)-1) * np]
  double *temp = V.f[F_temp] + (size_t)nn * np, *saln = V.f[F_saln] + (size_t)nn * np;
  double *dp = V.f[F_dp] + (size_t)nn * np, *sigma = V.f[F_sigma] + (size_t)nn * np;
  int kmax = 1;
  for (int k = 1; k <= kk; k++) {                                                    // :114-144
    const size_t o = c + (size_t)(k - 1) * np;
    const double d = dp[o];
    AR(D_TTEM, k) = temp[o];
    AR(D_SSAL, k) = saln[o];
    AR(D_DELP, k) = d;
    AR(D_DENS, k) = sigma[o];
    AR(D_NU, k) = V.f[F_difdia][o];
    for (int nt = 0; nt < ntr; nt++) AR(D_TTRC + nt, k) = V.f[F_trc][o + (size_t)(nn + nt * 2 * kk) * np];
    if (k >= 2 && d > EPSILP) kmax = k;
  }
  const int kfpl = V.m[I_kfpla][c + (size_t)(n - 1) * np];
  const int kmin = kfpl - 2;
  if (kmin < kmax) {
    // rstdns(k) is .false. only for k = kfpl and possibly kfpl+1, :150-155
    int rst1 = kfpl, rst2 = -1;
    if (kfpl != kmax)
      if (AR(D_DENS, kfpl) > .5 * (SIGR(kfpl) + SIGR(kfpl + 1))) rst2 = kfpl + 1;
    AR(D_DELP, kmin + 1) = AR(D_DELP, 2); AR(D_DELP, kmin) = AR(D_DELP, 1);        // :159-172
    AR(D_TTEM, kmin + 1) = AR(D_TTEM, 2); AR(D_TTEM, kmin) = AR(D_TTEM, 1);
    AR(D_SSAL, kmin + 1) = AR(D_SSAL, 2); AR(D_SSAL, kmin) = AR(D_SSAL, 1);
    AR(D_NU, kmin + 1) = AR(D_NU, 2); AR(D_NU, kmin) = AR(D_NU, 1);
    for (int nt = 0; nt < ntr; nt++) { AR(D_TTRC + nt, kmin + 1) = AR(D_TTRC + nt, 2); AR(D_TTRC + nt, kmin) = AR(D_TTRC + nt, 1); }
    PRES(kmin) = 0.;                                                               // :175-178
    for (int k = kmin; k <= kmax; k++) PRES(k + 1) = PRES(k) + AR(D_DELP, k);
    const double presb = PRES(kmax + 1);
    int k = kmin;                                                                  // :182-193
    AR(D_FPU, k) = 0.;
    {
      const double d0 = AR(D_DELP, k), d1 = AR(D_DELP, k + 1);
      AR(D_FPL, k) = fmin2(fmin2(PRES(k + 1), presb - PRES(k + 1)), cc * AR(D_NU, k) * (d0 + d1) / (2. * d0 * d1));
    }
    k = kmin + 1;
    AR(D_FPU, k) = AR(D_FPL, k - 1);
    {
      const double delpu = fmax2(ONEM, AR(D_DELP, k)), delpl = fmax2(ONEM, AR(D_DELP, k + 1));
      AR(D_FPL, k) = fmin2(fmin2(PRES(k + 1), presb - PRES(k + 1)), cc * AR(D_NU, k) * (delpu + delpl) / (2. * delpu * delpl));
    }
    AR(D_FPL, kmax) = 0.;
    if (kfpl <= kmax) {
      if (kfpl < kmax) {                                                           // :197-209
        k = kmax - 1;
        const double us = V.f[F_ustarb][c];
        const double nubbl = gbbl * (us * us * us) *
                             exp_libm(-(AR(D_DELP, k + 1) + .5 * AR(D_DELP, k)) * fabs(V.f[F_coriop][c]) * ALPHA0 /
                                 (kappa * fmax2(ustmin, us) * GRAV)) /
                             (ALPHA0 * GRAV * (SIGR(k + 1) - SIGR(k)));
        const double nuk = fmax2(AR(D_NU, k), nubbl);
        AR(D_NU, k) = nuk;
        V.f[F_difdia][c + (size_t)(k - 1) * np] = nuk;
      }
      k = kfpl - 1;                                                                // :217-274
      AR(D_DSGLI, k) = 1.;
      AR(D_FCL, k) = -AR(D_FPL, k);
      for (k = kfpl; k <= kmax - 1; k++) {
        if (k != rst1 && k != rst2) {
          const double tk = AR(D_TTEM, k), sk = AR(D_SSAL, k), dk = AR(D_DELP, k), nuk = AR(D_NU, k);
          const double dsgdt = eosd::dsigdt(P, tk, sk), dsgds = eosd::dsigds(P, tk, sk);
          const double su = fmax2(dsgmnr * (SIGR(k) - SIGR(k - 1)), dsgdt * (tk - AR(D_TTEM, k - 1)) + dsgds * (sk - AR(D_SSAL, k - 1)));
          const double sl = fmax2(dsgmnr * (SIGR(k + 1) - SIGR(k)), dsgdt * (AR(D_TTEM, k + 1) - tk) + dsgds * (AR(D_SSAL, k + 1) - sk));
          const double shm = 2. * su * sl / (su + sl);
          const double sg = .5 * (su + sl);
          const double sui = 1. / su, sli = 1. / sl;
          AR(D_DSGU, k) = su; AR(D_DSGL, k) = sl; AR(D_DSGHM, k) = shm; AR(D_DSG, k) = sg; AR(D_DSGUI, k) = sui; AR(D_DSGLI, k) = sli;
          const double fcmx = .25 * (sqrt(dk * dk + 4. * cc * nuk * sg * (sui + sli)) - dk) * shm * fcmxr;
          const double densk = AR(D_DENS, k);
          const double dsgc = densk - SIGR(k);
          double fcu = 0., fcl = 0.;
          if (dsgc > 0.) {
            if (AR(D_DENS, k - 1) < SIGR(k)) {
              double q = fmax2(0., (densk - SIGR(k + 1)) / ((SIGR(k) - SIGR(k + 1)) * (1. - dsgcr0)));
              q = fmax2(0., 1. - q * q);
              q = q * q * q;
              fcu = dsgc * dk;
              fcu = fmin2(q * fcu + (1. - q) * fcmx, fcu);
            }
          } else {
            if (AR(D_DENS, k + 1) > SIGR(k)) {
              double q = fmax2(0., (densk - SIGR(k - 1)) / ((SIGR(k) - SIGR(k - 1)) * (1. - dsgcr0)));
              q = fmax2(0., 1. - q * q);
              q = q * q * q;
              fcl = dsgc * dk;
              fcl = fmax2(q * fcl - (1. - q) * fcmx, fcl);
            }
          }
          AR(D_FCU, k) = fcu;
          AR(D_FCL, k) = fcl;
        } else {
          AR(D_DSGU, k) = 1.; AR(D_DSGL, k) = 1.; AR(D_DSGHM, k) = 1.; AR(D_DSG, k) = 1.; AR(D_DSGUI, k) = 1.; AR(D_DSGLI, k) = 1.;
          AR(D_FCL, k) = 0.; AR(D_FCU, k) = 0.;
        }
      }
      k = kmax;                                                                    // :275-287
      {
        const double tk = AR(D_TTEM, k), sk = AR(D_SSAL, k);
        const double dsgdt = eosd::dsigdt(P, tk, sk), dsgds = eosd::dsigds(P, tk, sk);
        const double su = fmax2(dsgmnr * (SIGR(k) - SIGR(k - 1)), dsgdt * (tk - AR(D_TTEM, k - 1)) + dsgds * (sk - AR(D_SSAL, k - 1)));
        const double sui = 1. / su;
        AR(D_DSGU, k) = su;
        AR(D_DSGUI, k) = sui;
        double fpu;
        if (AR(D_DENS, k) > SIGR(k) && AR(D_DENS, k - 1) < SIGR(k)) fpu = fmin2(AR(D_DELP, k - 1), (AR(D_DENS, k) - SIGR(k)) * AR(D_DELP, k) * sui);
        else fpu = 0.;
        AR(D_FPU, k) = fpu;
        AR(D_FCU, k) = fpu * su;
      }
      AR(D_FMAX, kfpl - 1) = 0.;                                                   // :292-330
      AR(D_FMAX, kmax) = 0.;
      bool done = false;
      int niter = 0, kfmaxu = 0;
      const double preskf = PRES(kfpl);
      while (!done) {
        done = true;
        for (k = kmax - 1; k >= kfpl; k--) {
          const double q = ((AR(D_FMAX, k + 1) + AR(D_FCU, k + 1)) * AR(D_DSGUI, k + 1) + presb - PRES(k + 1)) * AR(D_DSGL, k);
          const double fcl = fmax2(-q, AR(D_FCL, k));
          AR(D_FCL, k) = fcl;
          AR(D_FMAX, k) = q + fcl;
        }
        kfmaxu = 0;
        for (k = kfpl; k <= kmax - 1; k++) {
          const double q = ((AR(D_FMAX, k - 1) - AR(D_FCL, k - 1)) * AR(D_DSGLI, k - 1) + PRES(k) - preskf) * AR(D_DSGU, k);
          double fcu = AR(D_FCU, k);
          if (fcu > q) { fcu = q; AR(D_FCU, k) = q; done = false; }
          if (AR(D_FMAX, k) > q - fcu) { AR(D_FMAX, k) = q - fcu; kfmaxu = k; }
        }
        // the reference tests niter == 100 without ever incrementing niter in this loop (:317),
        // i.e. it never aborts here; we bound the loop defensively and flag it.
        if (++niter > 100000) { atomicOr(errflag, 1); break; }
      }
      k = kfpl - 1;                                                                // :334-353
      AR(D_F0, k) = 0.; AR(D_F, k) = 0.; AR(D_GTD, k) = 0.;
      double dflim = 0.;
      for (k = kfpl; k <= kmax - 1; k++) {
        const double nuk = AR(D_NU, k), sg = AR(D_DSG, k), sui = AR(D_DSGUI, k), sli = AR(D_DSGLI, k), fmx = AR(D_FMAX, k);
        const double fk = fmin2(fmin2(fmx, .5 * sqrt(cc * nuk * sg * (sui + sli)) * AR(D_DSGHM, k)), cc * nuk * sg / fmax2(EPSILP, AR(D_DELP, k)));
        AR(D_F, k) = fk;
        AR(D_FOLD, k) = fk;
        AR(D_H, k) = AR(D_FCU, k) * sui - AR(D_FCL, k) * sli + AR(D_FCL, k - 1) * AR(D_DSGLI, k - 1) - AR(D_FCU, k + 1) * AR(D_DSGUI, k + 1);
        dflim = fmax2(dflim, fmx);
      }
      k = kmax;
      AR(D_F0, k) = 0.; AR(D_F, k) = 0.; AR(D_GTD, k) = 0.;
      dflim = dflim * dfeps;
      niter = 0;                                                                   // :357-533
      bool dwnwrd = false;
      for (;;) {
        dwnwrd = !dwnwrd;
        double maxdf = 0., ctd, atd, bitd;
        bool remfmx = false;
        if (dwnwrd) {
          ctd = 0.; bitd = 1.;
          for (k = kfpl; k <= kmax - 1; k++) {
            const double fmx = AR(D_FMAX, k);
            if (remfmx) { AR(D_GTD, k) = 0.; AR(D_F0, k) = fmx; AR(D_F, k) = fmx; }
            else {
              const double sui = AR(D_DSGUI, k), sli = AR(D_DSGLI, k);
              const double f0m = AR(D_F0, k - 1), slim = AR(D_DSGLI, k - 1), fp = AR(D_F, k + 1), suip = AR(D_DSGUI, k + 1);
              const double q = f0m * slim + fp * suip - AR(D_DELP, k) - AR(D_H, k);
              const double r = 4. * cc * AR(D_NU, k) * AR(D_DSG, k) * (sui + sli);
              const double t = .25 * AR(D_DSGHM, k);
              double f0, dfdg;
              flux_solution(q, r, t, f0, dfdg);
              if (f0 >= fmx) { f0 = fmx; dfdg = 0.; if (k > kfmaxu) remfmx = true; }
              AR(D_F0, k) = f0;
              const double g = ctd * bitd;
              AR(D_GTD, k) = g;
              atd = -dfdg * slim;
              ctd = -dfdg * suip;
              bitd = 1. / (1. - atd * g);
              AR(D_F, k) = (f0 - atd * (AR(D_F, k - 1) - f0m) + ctd * fp) * bitd;
            }
          }
          for (k = kmax - 1; k >= kfpl; k--) {
            const double fk = fmin2(AR(D_FMAX, k), AR(D_F, k) - AR(D_GTD, k + 1) * AR(D_F, k + 1));
            AR(D_F, k) = fk;
            maxdf = fmax2(maxdf, fabs(fk - AR(D_FOLD, k)));
            AR(D_FOLD, k) = fk;
          }
        } else {
          atd = 0.; bitd = 1.;
          for (k = kmax - 1; k >= kfpl; k--) {
            const double fmx = AR(D_FMAX, k);
            if (remfmx) { AR(D_GTD, k) = 0.; AR(D_F0, k) = fmx; AR(D_F, k) = fmx; }
            else {
              const double sui = AR(D_DSGUI, k), sli = AR(D_DSGLI, k);
              const double fm = AR(D_F, k - 1), slim = AR(D_DSGLI, k - 1), f0p = AR(D_F0, k + 1), suip = AR(D_DSGUI, k + 1);
              const double q = fm * slim + f0p * suip - AR(D_DELP, k) - AR(D_H, k);
              const double r = 4. * cc * AR(D_NU, k) * AR(D_DSG, k) * (sui + sli);
              const double t = .25 * AR(D_DSGHM, k);
              double f0, dfdg;
              flux_solution(q, r, t, f0, dfdg);
              if (f0 >= fmx) { f0 = fmx; dfdg = 0.; if (k <= kfmaxu) remfmx = true; }
              AR(D_F0, k) = f0;
              const double g = atd * bitd;
              AR(D_GTD, k) = g;
              atd = -dfdg * slim;
              ctd = -dfdg * suip;
              bitd = 1. / (1. - ctd * g);
              AR(D_F, k) = (f0 + atd * fm - ctd * (AR(D_F, k + 1) - f0p)) * bitd;
            }
          }
          for (k = kfpl; k <= kmax - 1; k++) {
            const double fk = fmin2(AR(D_FMAX, k), AR(D_F, k) - AR(D_GTD, k - 1) * AR(D_F, k - 1));
            AR(D_F, k) = fk;
            maxdf = fmax2(maxdf, fabs(fk - AR(D_FOLD, k)));
            AR(D_FOLD, k) = fk;
          }
        }
        niter = niter + 1;
        if (maxdf <= dflim) break;
        if (niter == 100) { atomicOr(errflag, 2); break; }                         // :520-532 (xchalt)
      }
      for (k = kfpl; k <= kmax - 1; k++) {                                         // :536-540
        const double fk = AR(D_F, k);
        AR(D_FPU, k) = (fk + AR(D_FCU, k)) * AR(D_DSGUI, k);
        AR(D_FPL, k) = (fk - AR(D_FCL, k)) * AR(D_DSGLI, k);
      }
      AR(D_FPU, kfpl) = AR(D_FPL, kmin + 1);
    }
    {                                                                              // :546-576
      double ctd = 0., bitd = 1.;
      for (k = kmin; k <= kmax; k++) {
        const double g = ctd * bitd;
        AR(D_GTD, k) = g;
        const double dk = AR(D_DELP, k), fu = AR(D_FPU, k), fl = AR(D_FPL, k);
        const double q = 1. / (dk + fu + fl);
        const double atd = -fu * q;
        ctd = -fl * q;
        const double dtd = dk * q;
        bitd = 1. / (1. - atd * g);
        const int km1 = k - 1 > 1 ? k - 1 : 1;
        AR(D_SSAL, k) = (dtd * AR(D_SSAL, k) - atd * AR(D_SSAL, km1)) * bitd;
        AR(D_TTEM, k) = (dtd * AR(D_TTEM, k) - atd * AR(D_TTEM, km1)) * bitd;
        for (int nt = 0; nt < ntr; nt++) AR(D_TTRC + nt, k) = (dtd * AR(D_TTRC + nt, k) - atd * AR(D_TTRC + nt, km1)) * bitd;
      }
      for (k = kmax - 1; k >= kmin; k--) {
        const double g = AR(D_GTD, k + 1);
        const double sk = AR(D_SSAL, k) - g * AR(D_SSAL, k + 1);
        const double tk = AR(D_TTEM, k) - g * AR(D_TTEM, k + 1);
        AR(D_SSAL, k) = sk;
        AR(D_TTEM, k) = tk;
        AR(D_DENS, k) = eos::sig(P, tk, sk);
        for (int nt = 0; nt < ntr; nt++) AR(D_TTRC + nt, k) = AR(D_TTRC + nt, k) - g * AR(D_TTRC + nt, k + 1);
      }
      for (k = kfpl; k <= kmax - 1; k++)
        AR(D_DELP, k) = fmax2(0., AR(D_DELP, k) + AR(D_FPU, k) + AR(D_FPL, k) - AR(D_FPL, k - 1) - AR(D_FPU, k + 1));
      AR(D_DELP, kmax) = fmax2(0., AR(D_DELP, kmax) + AR(D_FPU, kmax) - AR(D_FPL, kmax - 1));
    }
    AR(D_TTEM, 1) = AR(D_TTEM, kmin); AR(D_TTEM, 2) = AR(D_TTEM, kmin + 1);        // :580-599
    AR(D_SSAL, 1) = AR(D_SSAL, kmin); AR(D_SSAL, 2) = AR(D_SSAL, kmin + 1);
    AR(D_DENS, 1) = AR(D_DENS, kmin); AR(D_DENS, 2) = AR(D_DENS, kmin + 1);
    if (kmin > 1) {
      if (kmin == 2) { AR(D_DELP, 2) = AR(D_DELP, kmin + 1); AR(D_DELP, kmin + 1) = 0.; }
      else AR(D_DELP, kmin) = 0.;
    }
    for (int nt = 0; nt < ntr; nt++) { AR(D_TTRC + nt, 1) = AR(D_TTRC + nt, kmin); AR(D_TTRC + nt, 2) = AR(D_TTRC + nt, kmin + 1); }
  }
  if (kfpl > kmax) {                                                               // :605-651
    const double t2 = AR(D_TTEM, 2);
    for (int k = 3; k <= kk; k++) {
      const double tk = fmax2(t2, V.f[F_temmin][c + (size_t)(k - 1) * np]);
      AR(D_TTEM, k) = tk;
      AR(D_DENS, k) = SIGR(k);
      AR(D_SSAL, k) = eosd::sofsig(P, SIGR(k), tk);
      AR(D_DELP, k) = 0.;
      for (int nt = 0; nt < ntr; nt++) {                                           // :612-626
        const double v = AR(D_TTRC + nt, 2);
        AR(D_TTRC + nt, k) = (P.itrtke >= 1 && nt + 1 == P.itrtke) ? fmax2(v, TKE_MIN)
                           : (P.itrtke >= 1 && P.gls && nt + 1 == P.itrgls) ? fmax2(v, GLS_PSI_MIN) : v;
      }
    }
  } else {
    const double tf = AR(D_TTEM, kfpl);
    for (int k = 3; k <= kfpl - 1; k++) {
      AR(D_TTEM, k) = tf;
      AR(D_DENS, k) = SIGR(k);
      AR(D_SSAL, k) = eosd::sofsig(P, SIGR(k), tf);
      AR(D_DELP, k) = 0.;
      for (int nt = 0; nt < ntr; nt++) AR(D_TTRC + nt, k) = AR(D_TTRC + nt, kfpl);
    }
    const double tm = AR(D_TTEM, kmax);
    for (int k = kmax + 1; k <= kk; k++) {
      AR(D_TTEM, k) = tm;
      AR(D_DENS, k) = SIGR(k);
      AR(D_SSAL, k) = eosd::sofsig(P, SIGR(k), tm);
      for (int nt = 0; nt < ntr; nt++) AR(D_TTRC + nt, k) = AR(D_TTRC + nt, kmax);
    }
  }
  double pacc = V.f[F_p][c];
  for (int k = 1; k <= kk; k++) {                                                  // :654-678
    const size_t o = c + (size_t)(k - 1) * np;
    const double d = AR(D_DELP, k);
    temp[o] = AR(D_TTEM, k);
    saln[o] = AR(D_SSAL, k);
    dp[o] = d;
    sigma[o] = AR(D_DENS, k);
    pacc = pacc + d;
    V.f[F_p][c + (size_t)k * np] = pacc;
    for (int nt = 0; nt < ntr; nt++) {                                             // :662-677
      const double v = AR(D_TTRC + nt, k);
      V.f[F_trc][o + (size_t)(nn + nt * 2 * kk) * np] = (P.itrtke >= 1 && nt + 1 == P.itrtke) ? fmax2(v, TKE_MIN)
                                                     : (P.itrtke >= 1 && P.gls && nt + 1 == P.itrgls) ? fmax2(v, GLS_PSI_MIN) : v;
    }
  }
  V.f[F_util1][c] = (double)kmin;                                                  // :681-700, :718
  if (kmin < kmax) {
    const double fl = AR(D_FPL, kmin);
    for (int k = 1; k <= kmin; k++) { V.f[F_fpug][c + (size_t)(k - 1) * np] = fl; V.f[F_fplg][c + (size_t)(k - 1) * np] = fl; }
    for (int k = kmin + 1; k <= kmax; k++) { V.f[F_fpug][c + (size_t)(k - 1) * np] = AR(D_FPU, k); V.f[F_fplg][c + (size_t)(k - 1) * np] = AR(D_FPL, k); }
    for (int k = kmax + 1; k <= kk; k++) { V.f[F_fpug][c + (size_t)(k - 1) * np] = 0.; V.f[F_fplg][c + (size_t)(k - 1) * np] = 0.; }
  } else
    for (int k = 1; k <= kk; k++) { V.f[F_fpug][c + (size_t)(k - 1) * np] = 0.; V.f[F_fplg][c + (size_t)(k - 1) * np] = 0.; }
}

__global__ void k_diapfl_kming(const DevView *__restrict__ Vp) {                                // :725-733
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1 || !V.m[I_ip][c]) return;
  V.m[I_kming][c] = (int)rint(V.f[F_util1][c]);
}

// momentum mixing of one component (blockIdx.y = 0: u, 1: v), :740-852 / :856-966.
// As in the column pass, the reference's work arrays uc, delp (the mixed layer moved to positions
// kmin, kmin+1) are the velocity and thickness arrays themselves under the position -> layer map
// (kmin -> 1, kmin+1 -> 2), the interface fluxes fpu/fpl are evaluated where the tridiagonal sweep
// consumes them, and the forward-eliminated velocity is kept in place; only gtd needs a work plane.
enum { U_GTD = 0, U_NSLOT };

// fluxes through the upper interface of position k (fpu(k)) and, same interface seen from above,
// fpl(k-1): the average of the two neighbouring p-columns, clipped at the velocity-point bottom (:779-817)
// the same from the six values it reads: p, fpug of level k and fplg of level k-1 at the two p-columns (m: mns, c)
__device__ inline void diapfl_mom_flux_v(double pm, double fum, double flm, double pc, double fuc, double flc, double pzb,
                                         double &fpu_k, double &fpl_km1) {
  double fpum, fplm, fpup, fplp;
  double pnew = pm, fu = fum, fl = flm;
  double pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpum = fu; fplm = fl; } else { fpum = fu; fplm = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpum = fu - pold + pzb; fplm = fl; } else { fpum = .5 * (fu + fl); fplm = fpum; }
  }
  pnew = pc; fu = fuc; fl = flc;
  pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpup = fu; fplp = fl; } else { fpup = fu; fplp = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpup = fu - pold + pzb; fplp = fl; } else { fpup = .5 * (fu + fl); fplp = fpup; }
  }
  fpu_k = .5 * (fpum + fpup);
  fpl_km1 = .5 * (fplm + fplp);
}
__device__ inline void diapfl_mom_flux(const double *p, const double *fpug, const double *fplg, size_t mns, size_t c, size_t np,
                                       int k, double pzb, double &fpu_k, double &fpl_km1) {
  double fpum, fplm, fpup, fplp;
  const size_t o = (size_t)(k - 1) * np, om = (size_t)(k - 2) * np;
  double pnew = p[mns + o], fu = fpug[mns + o], fl = fplg[mns + om];
  double pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpum = fu; fplm = fl; } else { fpum = fu; fplm = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpum = fu - pold + pzb; fplm = fl; } else { fpum = .5 * (fu + fl); fplm = fpum; }
  }
  pnew = p[c + o]; fu = fpug[c + o]; fl = fplg[c + om];
  pold = pnew - fl + fu;
  if (pold <= pzb) {
    if (pnew <= pzb) { fpup = fu; fplp = fl; } else { fpup = fu; fplp = fl - pnew + pzb; }
  } else {
    if (pnew <= pzb) { fpup = fu - pold + pzb; fplp = fl; } else { fpup = .5 * (fu + fl); fplp = fpup; }
  }
  fpu_k = .5 * (fpum + fpup);
  fpl_km1 = .5 * (fplm + fplp);
}

__global__ void k_diapfl_momentum(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane, mns = isv ? c - V.ni : c - 1;
  const int kk = V.kk, sb = isv ? U_NSLOT : 0;
  double *vel = (isv ? V.f[F_v] : V.f[F_u]) + (size_t)nn * np;
  const double *dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  const double *p = V.f[F_p], *fpug = V.f[F_fpug], *fplg = V.f[F_fplg];
#define LV(a, k) (a)[c + (size_t)((k)-1) * np]
  const int kmin = min(V.m[I_kming][mns], V.m[I_kming][c]);
  int kmax = 1;
#define MU_ 4       /* levels whose loads a sweep keeps in flight */
  for (int k0 = 2; k0 <= kk; k0 += 2 * MU_) {
    double a0[2 * MU_];
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++) a0[u] = LV(dpz, k0 + u <= kk ? k0 + u : kk);
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++)
      if (k0 + u <= kk && a0[u] > 0.) kmax = k0 + u;
  }
  if (!(kmin < kmax)) return;
  const double pzb = (isv ? V.f[F_pv] : V.f[F_pu])[c + (size_t)kk * np];
  // forward elimination over positions kmin..kmax, :822-836; position -> layer: kmin -> 1, kmin+1 -> 2
  double ctd = 0., bitd = 1., g = 0., uprev = 0.;
  double fu = 0., fl_prev_unused = 0.;                        // fpu(kmin) = 0
  (void)fl_prev_unused;
  double fu_next = 0., fl;
  for (int k0 = kmin; k0 <= kmax; k0 += MU_) {
   double b0[MU_], b1[MU_], b2[MU_], b3[MU_], b4[MU_], b5[MU_], b6[MU_], b7[MU_];
#pragma unroll
   for (int u = 0; u < MU_; u++) {
     const int kq = k0 + u <= kmax ? k0 + u : kmax, kn = kq + 1 <= kk ? kq + 1 : kk;
     const int lq = kq == kmin ? 1 : (kq == kmin + 1 ? 2 : kq);
     const size_t o = (size_t)(kn - 1) * np, om = (size_t)(kn - 2) * np;
     b0[u] = p[mns + o]; b1[u] = fpug[mns + o]; b2[u] = fplg[mns + om];
     b3[u] = p[c + o]; b4[u] = fpug[c + o]; b5[u] = fplg[c + om];
     b6[u] = LV(dpz, lq); b7[u] = LV(vel, lq);
   }
#pragma unroll
   for (int u = 0; u < MU_; u++) {
    const int k = k0 + u;
    if (k > kmax) break;
    const int lay = k == kmin ? 1 : (k == kmin + 1 ? 2 : k);
    if (k < kmax) diapfl_mom_flux_v(b0[u], b1[u], b2[u], b3[u], b4[u], b5[u], pzb, fu_next, fl);    // fpu(k+1), fpl(k)
    else fl = 0.;                                                                         // fpl(kmax) = 0
    g = ctd * bitd;
    AR(sb + U_GTD, k) = g;
    const double dk = b6[u];
    const double q = 1. / (dk + fu + fl);
    const double atd = -fu * q;
    ctd = -fl * q;
    const double dtd = dk * q;
    bitd = 1. / (1. - atd * g);
    const double uk = b7[u];
    // km1 = max(k-1, kmin): at k = kmin the reference reads uc(kmin) itself, i.e. the unmodified value
    uprev = (dtd * uk - atd * (k == kmin ? uk : uprev)) * bitd;
    LV(vel, lay) = uprev;
    fu = fu_next;
   }
  }
  // back substitution, :838
  double unext = uprev, gnext = g;
  for (int k0 = kmax - 1; k0 >= kmin; k0 -= 2 * MU_) {
    double b0[2 * MU_], b1[2 * MU_];
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++) {
      const int kq = k0 - u >= kmin ? k0 - u : kmin;
      const int lq = kq == kmin ? 1 : (kq == kmin + 1 ? 2 : kq);
      b0[u] = LV(vel, lq); b1[u] = AR(sb + U_GTD, kq);
    }
#pragma unroll
    for (int u = 0; u < 2 * MU_; u++) {
      const int k = k0 - u;
      if (k < kmin) break;
      const int lay = k == kmin ? 1 : (k == kmin + 1 ? 2 : k);
      unext = b0[u] - gnext * unext;
      LV(vel, lay) = unext;
      gnext = b1[u];
    }
  }
  const double ub = uprev;                                    // uc(kmax): untouched by the back substitution
  for (int k = kmax + 1; k <= kk; k++)
    if (fmin2(p[mns + (size_t)(k - 1) * np], LV(p, k)) < pzb) LV(vel, k) = ub;
}

// dpu/dpv at the new level from the updated p, :971-1000 (u: i 1..ii+1, j 1..jj ; v: i 1..ii, j 1..jj+1)
__global__ void k_diapfl_dpudpv(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const int k = by_;
  const size_t np = V.nplane, o0 = (size_t)k * np, o1 = (size_t)(k + 1) * np, ob = (size_t)V.kk * np;
  const double *p = V.f[F_p];
  if (V.m[I_iu][c] && j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1) {
    const size_t w = c - 1;
    const double q = fmin2(p[c + ob], p[w + ob]);
    V.f[F_dpu][c + (size_t)(k + nn) * np] = .5 * ((fmin2(q, p[w + o1]) - fmin2(q, p[w + o0])) + (fmin2(q, p[c + o1]) - fmin2(q, p[c + o0])));
  }
  if (V.m[I_iv][c] && j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii) {
    const size_t s = c - V.ni;
    const double q = fmin2(p[c + ob], p[s + ob]);
    V.f[F_dpv][c + (size_t)(k + nn) * np] = .5 * ((fmin2(q, p[s + o1]) - fmin2(q, p[s + o0])) + (fmin2(q, p[c + o1]) - fmin2(q, p[c + o0])));
  }
}

int st_diapfl(blomgpu_ctx *c, int n, int nn, int k1n) {
  (void)k1n;
  const DevView &h = c->h;
  if (h.ntr > MAXTR || D_TTRC + h.ntr > h.nwk) return ctx_fail(c, "diapfl: device work space too small");
  if (h.P.vcoord_tag != 1) return ctx_fail(c, "diapfl is only called for isopyc_bulkml (phy/mod_blom_step.F90:172-186)");
  if (int rc = ctx_err_words(c)) return rc;
  int *errflag = c->err_dev + 0;
  {
    TimeScope ts(c, "diapfl");
    if (c->diapfl_v == 3) { if (int rc = diapfl_column3_launch(c, n, nn, errflag)) return rc; }
    else if (c->diapfl_v == 2) { if (int rc = diapfl_column2_launch(c, n, nn, errflag)) return rc; }
    else hipLaunchKernelGGL(k_diapfl_column, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, n, nn, errflag);
    {                                                                                     // :711-713, :723
      double *ptrs[4] = {h.f[F_p], h.f[F_fpug], h.f[F_fplg], h.f[F_util1]};
      const int nl[4] = {h.kk + 1, h.kk, h.kk, 1}, it[4] = {1, 1, 1, 1};
      if (int rc = st_xctilr_multi(c, 4, ptrs, nl, 1, 1, it)) return rc;
    }
    hipLaunchKernelGGL(k_diapfl_kming, plane_grid(h), dim3(256), 0, c->stream, c->d);
    hipLaunchKernelGGL(k_diapfl_momentum, plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, nn);
    hipLaunchKernelGGL(k_diapfl_dpudpv, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
  }
  HIPCHK(c, hipGetLastError());
  // the reference aborts (xchalt) when the implicit solve does not converge, :520-530
  if (!c->defer_checks) return ctx_check_errors(c);
  return 0;
}
