/* TEST INFRASTRUCTURE (oracle): restatement of xctilr (single tile), init_fluxes, the
 * time-smoothing stages and the dp/dpu/dpv tail of mxlayr. */
#include "ostate.h"

/* xctilr, serial non-arctic form, phy/mod_xc.F90:4374-4419 */
void orc_xctilr(OState *S, double *a, int l1, int ld, int mh, int nh, int itype) {
  const int ii = S->ii, jj = S->jj;
  const int mhl = imax2(0, imin2(mh, NBDY)), nhl = imax2(0, imin2(nh, NBDY));
#define A(i, j, k) a[IX(S, i, j) + (size_t)S->nplane * ((k)-1)]
  if (S->nreg == 2) { /* arctic patch, phy/mod_xc.F90:4262-4372 */
    const double sgn = itype > 10 ? -1. : 1.;
    const int g = itype % 10;
    for (int k = l1; k <= ld; k++) {
      for (int j = 1; j <= nhl; j++)
        for (int i = 1; i <= ii; i++) A(i, 1 - j, k) = S->vland;
      if (g == 1 || g == 3) { /* p-, u-grid */
        for (int j = 0; j <= nhl; j++)
          for (int i = 1; i <= ii; i++) {
            const int io = g == 1 ? ii - (i - 1) % ii : (ii - (i - 1)) % ii + 1;
            A(i, jj + j, k) = sgn * A(io, jj - 1 - j, k);
          }
      } else { /* q-, v-grid */
        for (int i = ii / 2 + 1; i <= ii; i++) {
          const int io = g == 2 ? (ii - (i - 1)) % ii + 1 : ii - (i - 1) % ii;
          A(i, jj, k) = sgn * A(io, jj, k);
        }
        for (int j = 1; j <= nhl; j++)
          for (int i = 1; i <= ii; i++) {
            const int io = g == 2 ? (ii - (i - 1)) % ii + 1 : ii - (i - 1) % ii;
            A(i, jj + j, k) = sgn * A(io, jj - j, k);
          }
      }
    }
    if (mhl > 0) /* the reference runs this loop over k = 1..ld; callers here always pass l1 = 1 or own the levels */
      for (int k = l1; k <= ld; k++)
        for (int j = 1 - nhl; j <= jj + nhl; j++)
          for (int i = 1; i <= mhl; i++) { A(1 - i, j, k) = A(ii + 1 - i, j, k); A(ii + i, j, k) = A(i, j, k); }
    return;
  }
  if (nhl > 0) {
    if (S->nreg <= 2) {
      for (int k = l1; k <= ld; k++)
        for (int j = 1; j <= nhl; j++)
          for (int i = 1; i <= ii; i++) { A(i, 1 - j, k) = S->vland; A(i, jj + j, k) = S->vland; }
    } else {
      for (int k = l1; k <= ld; k++)
        for (int j = 1; j <= nhl; j++)
          for (int i = 1; i <= ii; i++) { A(i, 1 - j, k) = A(i, jj + 1 - j, k); A(i, jj + j, k) = A(i, j, k); }
    }
  }
  if (mhl > 0) {
    if (S->nreg == 0 || S->nreg == 4) {
      for (int k = l1; k <= ld; k++)
        for (int j = 1 - nhl; j <= jj + nhl; j++)
          for (int i = 1; i <= mhl; i++) { A(1 - i, j, k) = S->vland; A(ii + i, j, k) = S->vland; }
    } else {
      for (int k = l1; k <= ld; k++)
        for (int j = 1 - nhl; j <= jj + nhl; j++)
          for (int i = 1; i <= mhl; i++) { A(1 - i, j, k) = A(ii + 1 - i, j, k); A(ii + i, j, k) = A(i, j, k); }
    }
  }
#undef A
}

/* init_fluxes, phy/mod_state.F90:341-383 (update_flux_halos = .false.) */
void orc_init_fluxes(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)nn; (void)k1m; (void)k1n;
  for (int j = 0; j <= S->jj + 2; j++)
    for (int k = 1; k <= S->kk; k++) {
      int km = k + mm;
      for (int i = 0; i <= S->ii + 2; i++) {
        if (A2(S, iu, i, j)) { A3(S, uflx, i, j, km) = 0.; A3(S, utflx, i, j, km) = 0.; A3(S, usflx, i, j, km) = 0.; }
        if (A2(S, iv, i, j)) { A3(S, vflx, i, j, km) = 0.; A3(S, vtflx, i, j, km) = 0.; A3(S, vsflx, i, j, km) = 0.; }
      }
    }
}

/* initms, phy/mod_tmsmt.F90:161-205 */
void orc_initms(OState *S, int mm) {
  for (int j = 1; j <= S->jj; j++)
    for (int k = 1; k <= S->kk; k++) {
      int km = k + mm;
      for (int i = 1; i <= S->ii; i++)
        if (A2(S, ip, i, j)) {
          A3(S, told, i, j, k) = A3(S, temp, i, j, km);
          A3(S, sold, i, j, k) = A3(S, saln, i, j, km);
          for (int nt = 1; nt <= S->ntr; nt++) TRCOLD(S, i, j, k, nt) = TRC(S, i, j, km, nt);
        }
    }
}

/* tmsmt1, phy/mod_tmsmt.F90:209-277 */
void orc_tmsmt1(OState *S, int nn) {
  for (int j = 1; j <= S->jj; j++)
    for (int k = 1; k <= S->kk; k++) {
      int kn = k + nn;
      for (int i = 1; i <= S->ii; i++) {
        if (A2(S, ip, i, j)) {
          A3(S, dpold, i, j, kn) = A3(S, dp, i, j, kn);
          A3(S, told, i, j, k) = A3(S, temp, i, j, kn);
          A3(S, sold, i, j, k) = A3(S, saln, i, j, kn);
          for (int nt = 1; nt <= S->ntr; nt++) TRCOLD(S, i, j, k, nt) = TRC(S, i, j, kn, nt);
        }
        if (S->vcoord_tag == 1) {
          if (A2(S, iu, i, j)) A3(S, dpuold, i, j, k) = A3(S, dpu, i, j, kn);
          if (A2(S, iv, i, j)) A3(S, dpvold, i, j, k) = A3(S, dpv, i, j, kn);
        }
      }
    }
}

/* p(k+1)=p(k)+dp(k+off) over -2..+2 and dpu/dpv over -1..+2:
 * phy/mod_tmsmt.F90:354-391 == phy/mod_mxlayr.F90:1270-1310 == phy/mod_pgforc.F90:451-485 */
static void p_dpu_dpv(OState *S, int off, int with_pupv) {
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  for (int j = -2; j <= jj + 2; j++)
    for (int k = 1; k <= kk; k++)
      for (int i = -2; i <= ii + 2; i++)
        if (A2(S, ip, i, j)) A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, k + off);
  for (int j = -1; j <= jj + 2; j++)
    for (int k = 1; k <= kk; k++) {
      int kx = k + off;
      for (int i = -1; i <= ii + 2; i++) {
        if (A2(S, iu, i, j)) {
          double q = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i - 1, j, kk + 1));
          A3(S, dpu, i, j, kx) = .5 * ((fmin2(q, A3(S, p, i - 1, j, k + 1)) - fmin2(q, A3(S, p, i - 1, j, k))) +
                                       (fmin2(q, A3(S, p, i, j, k + 1)) - fmin2(q, A3(S, p, i, j, k))));
          if (with_pupv) A3(S, pu, i, j, k + 1) = A3(S, pu, i, j, k) + A3(S, dpu, i, j, kx);
        }
        if (A2(S, iv, i, j)) {
          double q = fmin2(A3(S, p, i, j, kk + 1), A3(S, p, i, j - 1, kk + 1));
          A3(S, dpv, i, j, kx) = .5 * ((fmin2(q, A3(S, p, i, j - 1, k + 1)) - fmin2(q, A3(S, p, i, j - 1, k))) +
                                       (fmin2(q, A3(S, p, i, j, k + 1)) - fmin2(q, A3(S, p, i, j, k))));
          if (with_pupv) A3(S, pv, i, j, k + 1) = A3(S, pv, i, j, k) + A3(S, dpv, i, j, kx);
        }
      }
    }
}
void orc_p_dpu_dpv(OState *S, int off, int with_pupv) { p_dpu_dpv(S, off, with_pupv); }

/* tmsmt2, phy/mod_tmsmt.F90:281-410 */
void orc_tmsmt2(OState *S, int m, int mm, int nn, int k1m) {
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  const double wts1 = S->wts1, wts2 = S->wts2;
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++) {
      if (!A2(S, ip, i, j)) continue;
      double pbfaco = 0., pbfacn = 0.;
      for (int k = 1; k <= kk; k++) {
        int kn = k + nn;
        pbfaco = pbfaco + A3(S, dpold, i, j, kn);
        pbfacn = pbfacn + A3(S, dp, i, j, kn);
      }
      pbfaco = A3(S, pb, i, j, m) / pbfaco;
      pbfacn = A3(S, pb, i, j, m) / pbfacn;
      for (int k = 1; k <= kk; k++) {
        int km = k + mm, kn = k + nn;
        double pold = fmax2(0., A3(S, dpold, i, j, kn) * pbfaco);
        double pmid = fmax2(0., A3(S, dp, i, j, km));
        double pnew = fmax2(0., A3(S, dp, i, j, kn) * pbfacn);
        A3(S, dp, i, j, km) = wts1 * pmid + wts2 * (pold + pnew);
        pold = pold + EPSILP;
        pmid = pmid + EPSILP;
        pnew = pnew + EPSILP;
        A3(S, temp, i, j, km) = (wts1 * pmid * A3(S, temp, i, j, km) +
                                 wts2 * (pold * A3(S, told, i, j, k) + pnew * A3(S, temp, i, j, kn))) /
                                (A3(S, dp, i, j, km) + EPSILP);
        A3(S, saln, i, j, km) = (wts1 * pmid * A3(S, saln, i, j, km) +
                                 wts2 * (pold * A3(S, sold, i, j, k) + pnew * A3(S, saln, i, j, kn))) /
                                (A3(S, dp, i, j, km) + EPSILP);
        for (int nt = 1; nt <= S->ntr; nt++)
          TRC(S, i, j, km, nt) = (wts1 * pmid * TRC(S, i, j, km, nt) +
                                  wts2 * (pold * TRCOLD(S, i, j, k, nt) + pnew * TRC(S, i, j, kn, nt))) /
                                 (A3(S, dp, i, j, km) + EPSILP);
      }
    }
  orc_xctilr(S, S->dp + (size_t)S->nplane * (k1m - 1), 1, kk, 3, 3, 1);
  if (S->vcoord_tag == 1) p_dpu_dpv(S, mm, 0);
  else
    for (int j = -2; j <= jj + 2; j++)
      for (int k = 1; k <= kk; k++)
        for (int i = -2; i <= ii + 2; i++)
          if (A2(S, ip, i, j)) A3(S, p, i, j, k + 1) = A3(S, p, i, j, k) + A3(S, dp, i, j, k + mm);
}

/* tail of mxlayr, phy/mod_mxlayr.F90:1266-1310 */
void orc_mxlayr_tail(OState *S, int nn, int k1n) {
  orc_xctilr(S, S->dp + (size_t)S->nplane * (k1n - 1), 1, S->kk, 3, 3, 1);
  p_dpu_dpv(S, nn, 0);
}

/* updtrc, trc/mod_tracers_update.F90:152-170: hamocc_step (iHAMOCC, not part of this path) + idlage_step,
 * idlage/mod_idlage.F90:57-96: the ideal age tracer is reset in the surface layer and aged by
 * delt1 / (86400 nday_in_year) below it, on all p-points including the halo. */
void orc_updtrc(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)mm; (void)k1m;
  if (S->itriag < 1 || S->itriag > S->ntr) return;
  const int kk = S->kk;
  const double q = S->delt1 / (86400. * S->nday_in_year);
  for (int j = 1 - NBDY; j <= S->jj + NBDY; j++)
    for (int i = 1 - NBDY; i <= S->ii + NBDY; i++) {
      if (!A2(S, ip, i, j)) continue;
      TRC(S, i, j, k1n, S->itriag) = 0.;
      for (int k = 2; k <= kk; k++) TRC(S, i, j, k + nn, S->itriag) = TRC(S, i, j, k + nn, S->itriag) + q;
    }
}

/* xcsum, phy/mod_xc.F90:4116-4161: masked sum of a 2-D array, reproducible bit for bit: every row is summed
 * in strips of 2*nbdy+1 = 9 points (strip sums added to the row sum in order), then the row sums are added
 * serially.  The mask of the global sums is ips (phy/mod_inigeo.F90:189-208): ip, without the seam row of an
 * arctic patch. */
double orc_xcsum(const OState *S, const double *a, const int *mask, int use_ips) {
  const int ii = S->ii, jj = S->jj;
  double total = 0.;
  for (int j = 1; j <= jj; j++) {
    double sum8 = 0.;
    for (int i1 = 1; i1 <= ii; i1 += 2 * NBDY + 1) {
      double sum8p = 0.;
      const int ie = i1 + 2 * NBDY < ii ? i1 + 2 * NBDY : ii;
      for (int i = i1; i <= ie; i++) {
        int mk = mask[IX(S, i, j)];
        if (use_ips && S->nreg == 2 && j >= jj) mk = 0;
        if (mk == 1) sum8p = sum8p + a[IX(S, i, j)];
      }
      sum8 = sum8 + sum8p;
    }
    total = j == 1 ? sum8 : total + sum8;
  }
  return total;
}

/* budget_sums, phy/mod_budget.F90:95-196 (use_TRC; no GLS): mass weighted column sums of salinity and
 * temperature into util1, util2 (and of the TKE tracer into util3) and their global sums; then the same for tracer 1 through util1.  The salt
 * correction term of call 5 (:182-194) needs mod_forcing's salt_corr, which no stage of this path produces:
 * it is taken as zero.  Results: S->budget[0..3][ncall-1][n-1] = sdp, tdp, trdp, tkedp. */
void orc_budget_sums(OState *S, int ncall, int n, int nn) {
  if (!S->cnsvdi) return;
  const int ii = S->ii, jj = S->jj, kk = S->kk;
  for (int j = 1; j <= jj; j++)
    for (int i = 1; i <= ii; i++)
      if (A2(S, ip, i, j)) { A2(S, util1, i, j) = 0.; A2(S, util2, i, j) = 0.; if (S->itrtke >= 1) A2(S, util3, i, j) = 0.; }
  for (int j = 1; j <= jj; j++)
    for (int k = 1; k <= kk; k++)
      for (int i = 1; i <= ii; i++)
        if (A2(S, ip, i, j)) {
          const double q = A3(S, dp, i, j, k + nn) * A2(S, scp2, i, j);
          A2(S, util1, i, j) = A2(S, util1, i, j) + A3(S, saln, i, j, k + nn) * q;
          A2(S, util2, i, j) = A2(S, util2, i, j) + A3(S, temp, i, j, k + nn) * q;
          if (S->itrtke >= 1) A2(S, util3, i, j) = A2(S, util3, i, j) + TRC(S, i, j, k + nn, S->itrtke) * q;
        }
  S->budget[0][ncall - 1][n - 1] = orc_xcsum(S, S->util1, S->ip, 1);
  S->budget[1][ncall - 1][n - 1] = orc_xcsum(S, S->util2, S->ip, 1);
  if (S->itrtke >= 1) S->budget[3][ncall - 1][n - 1] = orc_xcsum(S, S->util3, S->ip, 1);
  if (S->ntr >= 1) {
    for (int j = 1; j <= jj; j++)
      for (int i = 1; i <= ii; i++)
        if (A2(S, ip, i, j)) A2(S, util1, i, j) = 0.;
    for (int j = 1; j <= jj; j++)
      for (int k = 1; k <= kk; k++)
        for (int i = 1; i <= ii; i++)
          if (A2(S, ip, i, j)) {
            const double q = A3(S, dp, i, j, k + nn) * A2(S, scp2, i, j);
            A2(S, util1, i, j) = A2(S, util1, i, j) + TRC(S, i, j, k + nn, 1) * q;
          }
    S->budget[2][ncall - 1][n - 1] = orc_xcsum(S, S->util1, S->ip, 1);
  }
}
