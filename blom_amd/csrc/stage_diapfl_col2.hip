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

enum { E_SU, E_SL, E_FCU, E_FCL, E_FMAX, E_H, E_R, E_T, E_F, E_FT, E_GTD, E_NSLOT };
#define W(slot, k) w_##slot[c + (size_t)((k)-1) * np]
#define ST(a, k) (a)[c + (size_t)((k)-1) * np]
#define SIGR(k) sigr[c + (size_t)((k)-1) * np]
#define PRES(k) pres[c + (size_t)((k)-1) * np]
#define TRC(nt, k) trc[c + (size_t)((k)-1 + (nt)*2 * kk) * np]

// one wavefront per 64 columns and ~1700 wavefronts in all: occupancy cannot exceed 2 waves per SIMD anyway,
// so let the register allocator use up to 256 VGPRs instead of spilling at 128
__global__ void __launch_bounds__(64, 1) k_diapfl_column2(const DevView *__restrict__ Vp, int n, int nn, int *__restrict__ errflag) {
  const DevView &V = *Vp;
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
  const double *__restrict__ sigr = V.f[F_sigmar];
  double *__restrict__ pres = V.f[F_wkp1];
  // one restrict pointer per work plane: the planes do not overlap, and knowing it lets the compiler keep the loads
  // of the next levels in flight across the stores of this one
  double *__restrict__ w_E_SU = WK(V, E_SU), *__restrict__ w_E_SL = WK(V, E_SL), *__restrict__ w_E_FCU = WK(V, E_FCU);
  double *__restrict__ w_E_FCL = WK(V, E_FCL), *__restrict__ w_E_FMAX = WK(V, E_FMAX), *__restrict__ w_E_H = WK(V, E_H);
  double *__restrict__ w_E_R = WK(V, E_R), *__restrict__ w_E_T = WK(V, E_T), *__restrict__ w_E_F = WK(V, E_F);
  double *__restrict__ w_E_FT = WK(V, E_FT), *__restrict__ w_E_GTD = WK(V, E_GTD);
  double *__restrict__ temp = V.f[F_temp] + (size_t)nn * np, *__restrict__ saln = V.f[F_saln] + (size_t)nn * np;
  double *__restrict__ dp = V.f[F_dp] + (size_t)nn * np, *__restrict__ sigma = V.f[F_sigma] + (size_t)nn * np;
  double *__restrict__ trc = V.f[F_trc] + (size_t)nn * np, *__restrict__ nu = V.f[F_difdia];
  double *__restrict__ fpug = V.f[F_fpug], *__restrict__ fplg = V.f[F_fplg];

  int kmax = 1;                                                                      // :139-143
  for (int k = 2; k <= kk; k++)
    if (ST(dp, k) > EPSILP) kmax = k;
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
    for (int k = kfpl; k <= kmax; k++) { acc = acc + ST(dp, k); PRES(k + 1) = acc; }
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
      // ---- stratification and density-restoring fluxes, :217-287 ------------------------------
      const int rst1 = kfpl;
      double tm = ST(temp, 2), sm = ST(saln, 2);                    // position kfpl-1 holds layer 2
      double tk = ST(temp, kfpl), sk = ST(saln, kfpl);
      double dens_m = ST(sigma, kfpl - 1), dens_k = ST(sigma, kfpl);
      double sr_m = SIGR(kfpl - 1), sr_k = SIGR(kfpl);
      int rst2 = -1;
      if (kfpl != kmax)
        if (dens_k > .5 * (sr_k + SIGR(kfpl + 1))) rst2 = kfpl + 1;
      #pragma unroll 4
      for (k = kfpl; k <= kmax - 1; k++) {
        const double tp = ST(temp, k + 1), sp = ST(saln, k + 1), dens_p = ST(sigma, k + 1), sr_p = SIGR(k + 1);
        double su = 1., sl = 1., fcu = 0., fcl = 0.;
        if (k != rst1 && k != rst2) {
          const double dk = ST(dp, k), nuk = ST(nu, k);
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
        W(E_SU, k) = su; W(E_SL, k) = sl; W(E_FCU, k) = fcu; W(E_FCL, k) = fcl;
        tm = tk; sm = sk; tk = tp; sk = sp;
        dens_m = dens_k; dens_k = dens_p; sr_m = sr_k; sr_k = sr_p;
      }
      // k = kmax, :275-287 (tk, sk, dens_k, sr_k now belong to kmax; *_m to position kmax-1)
      double su_b, sui_b, fpu_b, fcu_b;
      {
        const double dsgdt = eosd::dsigdt(P, tk, sk), dsgds = eosd::dsigds(P, tk, sk);
        su_b = fmax2(dsgmnr * (sr_k - sr_m), dsgdt * (tk - tm) + dsgds * (sk - sm));
        sui_b = 1. / su_b;
        const double dkm1 = kmax - 1 == kmin + 1 ? d2 : ST(dp, kmax - 1);
        if (dens_k > sr_k && dens_m < sr_k) fpu_b = fmin2(dkm1, (dens_k - sr_k) * ST(dp, kmax) * sui_b);
        else fpu_b = 0.;
        fcu_b = fpu_b * su_b;
      }
      // ---- flux limiter, :292-330 ------------------------------------------------------------
      bool done = false;
      int niter = 0, kfmaxu = 0;
      while (!done) {
        done = true;
        double fmax_p = 0., fcu_p = fcu_b, sui_p = sui_b;
        #pragma unroll 4
        for (k = kmax - 1; k >= kfpl; k--) {
          const double q = ((fmax_p + fcu_p) * sui_p + presb - PRES(k + 1)) * W(E_SL, k);
          const double fcl = fmax2(-q, W(E_FCL, k));
          W(E_FCL, k) = fcl;
          fmax_p = q + fcl;
          W(E_FMAX, k) = fmax_p;
          fcu_p = W(E_FCU, k);
          sui_p = 1. / W(E_SU, k);
        }
        kfmaxu = 0;
        double fmax_m = 0., fcl_m = -fpl1, sli_m = 1.;
        #pragma unroll 4
        for (k = kfpl; k <= kmax - 1; k++) {
          const double q = ((fmax_m - fcl_m) * sli_m + PRES(k) - preskf) * W(E_SU, k);
          double fcu = W(E_FCU, k);
          if (fcu > q) { fcu = q; W(E_FCU, k) = q; done = false; }
          double fm = W(E_FMAX, k);
          if (fm > q - fcu) { fm = q - fcu; W(E_FMAX, k) = fm; kfmaxu = k; }
          fmax_m = fm;
          fcl_m = W(E_FCL, k);
          sli_m = 1. / W(E_SL, k);
        }
        // the reference tests niter == 100 without ever incrementing niter in this loop (:317),
        // i.e. it never aborts here; we bound the loop defensively and flag it.
        if (++niter > 100000) { atomicOr(errflag, 1); break; }
      }
      // ---- first guess, :334-353 ---------------------------------------------------------------
      double dflim = 0.;
      {
        double fcl_m = -fpl1, sli_m = 1.;
        double fcu_k = kfpl <= kmax - 1 ? W(E_FCU, kfpl) : 0., su_k = kfpl <= kmax - 1 ? W(E_SU, kfpl) : 1.;
        #pragma unroll 4
        for (k = kfpl; k <= kmax - 1; k++) {
          const double fcu_n = k + 1 <= kmax - 1 ? W(E_FCU, k + 1) : fcu_b;
          const double su_n = k + 1 <= kmax - 1 ? W(E_SU, k + 1) : su_b;
          const double nuk = ST(nu, k), sl = W(E_SL, k), fmx = W(E_FMAX, k), fcl_k = W(E_FCL, k);
          const double su = su_k;
          const double shm = 2. * su * sl / (su + sl);
          const double sg = .5 * (su + sl);
          const double sui = 1. / su, sli = 1. / sl;
          const double fk = fmin2(fmin2(fmx, .5 * sqrt(cc * nuk * sg * (sui + sli)) * shm), cc * nuk * sg / fmax2(EPSILP, ST(dp, k)));
          W(E_F, k) = fk;
          W(E_H, k) = fcu_k * sui - fcl_k * sli + fcl_m * sli_m - fcu_n * (1. / su_n);
          W(E_R, k) = 4. * cc * nuk * sg * (sui + sli);
          W(E_T, k) = .25 * shm;
          dflim = fmax2(dflim, fmx);
          fcl_m = fcl_k; sli_m = sli; fcu_k = fcu_n; su_k = su_n;
        }
      }
      dflim = dflim * dfeps;
      // ---- implicit solve by alternating sweeps, :357-533 ------------------------------------
      niter = 0;
      bool dwnwrd = false;
      for (;;) {
        dwnwrd = !dwnwrd;
        double maxdf = 0., ctd = 0., atd = 0., bitd = 1.;
        bool remfmx = false;
        if (dwnwrd) {
          double f0m = 0., fnew_m = 0., slim = 1.;             // f0, f, dsgli of level kfpl-1
          #pragma unroll 4
          for (k = kfpl; k <= kmax - 1; k++) {
            const double fmx = W(E_FMAX, k);
            if (remfmx) { W(E_GTD, k) = 0.; W(E_FT, k) = fmx; }
            else {
              const double fp = k + 1 <= kmax - 1 ? W(E_F, k + 1) : 0.;
              const double suip = k + 1 <= kmax - 1 ? 1. / W(E_SU, k + 1) : sui_b;
              const double q = f0m * slim + fp * suip - ST(dp, k) - W(E_H, k);
              double f0, dfdg;
              flux_solution(q, W(E_R, k), W(E_T, k), f0, dfdg);
              if (f0 >= fmx) { f0 = fmx; dfdg = 0.; if (k > kfmaxu) remfmx = true; }
              const double g = ctd * bitd;
              W(E_GTD, k) = g;
              atd = -dfdg * slim;
              ctd = -dfdg * suip;
              bitd = 1. / (1. - atd * g);
              fnew_m = (f0 - atd * (fnew_m - f0m) + ctd * fp) * bitd;
              W(E_FT, k) = fnew_m;
              f0m = f0;
              slim = 1. / W(E_SL, k);
            }
          }
          double fnew_p = 0., gtd_p = 0.;                       // f, gtd of level kmax
          #pragma unroll 4
          for (k = kmax - 1; k >= kfpl; k--) {
            const double fk = fmin2(W(E_FMAX, k), W(E_FT, k) - gtd_p * fnew_p);
            maxdf = fmax2(maxdf, fabs(fk - W(E_F, k)));
            W(E_F, k) = fk;
            fnew_p = fk;
            gtd_p = W(E_GTD, k);
          }
        } else {
          double f0p = 0., fnew_p = 0., suip = sui_b;           // f0, f of level kmax; dsgui(kmax)
          #pragma unroll 4
          for (k = kmax - 1; k >= kfpl; k--) {
            const double fmx = W(E_FMAX, k);
            if (remfmx) { W(E_GTD, k) = 0.; W(E_FT, k) = fmx; }
            else {
              const double fm = k - 1 >= kfpl ? W(E_F, k - 1) : 0.;
              const double slim = k - 1 >= kfpl ? 1. / W(E_SL, k - 1) : 1.;
              const double q = fm * slim + f0p * suip - ST(dp, k) - W(E_H, k);
              double f0, dfdg;
              flux_solution(q, W(E_R, k), W(E_T, k), f0, dfdg);
              if (f0 >= fmx) { f0 = fmx; dfdg = 0.; if (k <= kfmaxu) remfmx = true; }
              const double g = atd * bitd;
              W(E_GTD, k) = g;
              atd = -dfdg * slim;
              ctd = -dfdg * suip;
              bitd = 1. / (1. - ctd * g);
              fnew_p = (f0 + atd * fm - ctd * (fnew_p - f0p)) * bitd;
              W(E_FT, k) = fnew_p;
              f0p = f0;
              suip = 1. / W(E_SU, k);
            }
          }
          double fnew_m = 0., gtd_m = 0.;                       // f, gtd of level kfpl-1
          #pragma unroll 4
          for (k = kfpl; k <= kmax - 1; k++) {
            const double fk = fmin2(W(E_FMAX, k), W(E_FT, k) - gtd_m * fnew_m);
            maxdf = fmax2(maxdf, fabs(fk - W(E_F, k)));
            W(E_F, k) = fk;
            fnew_m = fk;
            gtd_m = W(E_GTD, k);
          }
        }
        niter = niter + 1;
        if (maxdf <= dflim) break;
        if (niter == 100) { atomicOr(errflag, 2); break; }                         // :520-532 (xchalt)
      }
      // ---- interface fluxes, :536-541 -------------------------------------------------------------
      #pragma unroll 4
      for (k = kfpl; k <= kmax - 1; k++) {
        const double fk = W(E_F, k);
        ST(fpug, k) = (fk + W(E_FCU, k)) * (1. / W(E_SU, k));
        ST(fplg, k) = (fk - W(E_FCL, k)) * (1. / W(E_SL, k));
      }
      ST(fpug, kmax) = fpu_b;
      ST(fplg, kmax) = 0.;
      ST(fpug, kfpl) = fpl1;                                                         // :541
    }
    // ---- implicit mixing of S, T, tracers over positions kmin..kmax, :546-576, fused with the layer
    //      thickness update :572-576 (which only reads fluxes) ------------------------------------
    {
      const bool interior = kfpl <= kmax;
      double ctd = 0., bitd = 1., g = 0.;
      const int km1 = kmin - 1 > 1 ? kmin - 1 : 1;
      double s_prev = ST(saln, km1), t_prev = ST(temp, km1);
      double tr_prev[MAXTR];
#pragma unroll
      for (int nt = 0; nt < MAXTR; nt++) tr_prev[nt] = nt < ntr ? TRC(nt, km1) : 0.;
      double fl_m = 0.;                                   // fpl of the previous position
      double fu_next = interior ? ST(fpug, kfpl) : 0.;
      for (int pos = kmin; pos <= kmax; pos++) {
        const int lay = pos == kmin ? 1 : (pos == kmin + 1 ? 2 : pos);
        double dk, fu, fl;
        if (pos == kmin) { dk = d1; fu = 0.; fl = fpl0; }
        else if (pos == kmin + 1) { dk = d2; fu = fpl0; fl = fpl1; }
        else { dk = ST(dp, pos); fu = fu_next; fl = ST(fplg, pos); }
        g = ctd * bitd;
        W(E_GTD, pos) = g;
        const double q = 1. / (dk + fu + fl);
        const double atd = -fu * q;
        ctd = -fl * q;
        const double dtd = dk * q;
        bitd = 1. / (1. - atd * g);
        s_prev = (dtd * ST(saln, lay) - atd * s_prev) * bitd;
        t_prev = (dtd * ST(temp, lay) - atd * t_prev) * bitd;
        ST(saln, lay) = s_prev;
        ST(temp, lay) = t_prev;
#pragma unroll
        for (int nt = 0; nt < MAXTR; nt++)
          if (nt < ntr) { tr_prev[nt] = (dtd * TRC(nt, lay) - atd * tr_prev[nt]) * bitd; TRC(nt, lay) = tr_prev[nt]; }
        if (pos >= kfpl) {                                // interior layers: new thickness
          if (pos < kmax) {
            fu_next = ST(fpug, pos + 1);
            ST(dp, pos) = fmax2(0., dk + fu + fl - fl_m - fu_next);
          } else
            ST(dp, pos) = fmax2(0., dk + fu - fl_m);
        } else if (pos == kmax && kmin <= 2)              // kmax = kmin+1: layer 2 keeps the update only
          ST(dp, 2) = fmax2(0., dk + fu - fl_m);          // when its position is copied back (:593-599)
        fl_m = fl;
      }
      double s_next = s_prev, t_next = t_prev, g_next = g;
      double tr_next[MAXTR];
#pragma unroll
      for (int nt = 0; nt < MAXTR; nt++) tr_next[nt] = tr_prev[nt];
      for (int pos = kmax - 1; pos >= kmin; pos--) {
        const int lay = pos == kmin ? 1 : (pos == kmin + 1 ? 2 : pos);
        s_next = ST(saln, lay) - g_next * s_next;
        t_next = ST(temp, lay) - g_next * t_next;
        ST(saln, lay) = s_next;
        ST(temp, lay) = t_next;
        ST(sigma, lay) = eos::sig(P, t_next, s_next);
#pragma unroll
        for (int nt = 0; nt < MAXTR; nt++)
          if (nt < ntr) { tr_next[nt] = TRC(nt, lay) - g_next * tr_next[nt]; TRC(nt, lay) = tr_next[nt]; }
        g_next = W(E_GTD, pos);
      }
    }
    // dens is the one work array the reference does not move with the mixed layer (:159-172): when the
    // column has no interior mass (kmax = kmin+1) position kmin+1 is not re-evaluated by the sweep
    // above and the copy-back :580-599 hands layer 2 the old density of layer kmin+1
    if (kmax == kmin + 1 && kmin >= 2) ST(sigma, 2) = ST(sigma, kmin + 1);
  }
  // ---- massless layers, :605-651 ---------------------------------------------------------------
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
    for (int k = 3; k <= kfpl - 1; k++) {
      ST(temp, k) = tf;
      ST(sigma, k) = SIGR(k);
      ST(saln, k) = eosd::sofsig(P, SIGR(k), tf);
      ST(dp, k) = 0.;
      for (int nt = 0; nt < ntr; nt++) TRC(nt, k) = TRC(nt, kfpl);
    }
    const double tm = ST(temp, kmax);
    for (int k = kmax + 1; k <= kk; k++) {
      ST(temp, k) = tm;
      ST(sigma, k) = SIGR(k);
      ST(saln, k) = eosd::sofsig(P, SIGR(k), tm);
      for (int nt = 0; nt < ntr; nt++) TRC(nt, k) = TRC(nt, kmax);
    }
  }
  // ---- lower bounds of the TKE / generic-length-scale tracers (:612-626 in the massless layers, :662-677 on
  //      every layer at the copy-back; max is idempotent, so one pass over the column covers both) ----------
  if (P.itrtke >= 1) {
    const int a = P.itrtke - 1, b = P.itrgls - 1;
    for (int k = 1; k <= kk; k++) {
      if (a < ntr) TRC(a, k) = fmax2(TRC(a, k), TKE_MIN);
      if (P.gls && b >= 0 && b < ntr) TRC(b, k) = fmax2(TRC(b, k), GLS_PSI_MIN);
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
}

int diapfl_column2_launch(blomgpu_ctx *c, int n, int nn, int *errflag) {
  const DevView &h = c->h;
  if (h.ntr > MAXTR || E_NSLOT > h.nwk) return ctx_fail(c, "diapfl: more than 4 tracers / work space too small");
  hipLaunchKernelGGL(k_diapfl_column2, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, n, nn, errflag);
  return 0;
}
