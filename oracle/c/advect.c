/* TEST INFRASTRUCTURE (oracle): restatement of advect (phy/mod_advect.F90:59-189, advmth =
 * 'remap') and of the incremental remapping remap/triint/penint (phy/mod_remap.F90:53-199,
 * :205-1522) for use_TRC = .true., use_ATRC = .false., no TKE tracers.
 *
 * The four directional variants of the u-face (and of the v-face) flux polygons in the
 * Fortran differ only in the donor cell and in the sign of the half-cell offset; they are
 * written once here with s = +/-1 (x + s*.5 with s*.5 exact reproduces x+.5 / x-.5). */
#include "ostate.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define DPEPS 1.e-12 /* phy/mod_remap.F90:40 */

/* triint, phy/mod_remap.F90:53-102 */
static void triint(double ac, double x1, double y1, double x2, double y2, double x3, double y3, double *a,
                   double *ax, double *ay, double *axx, double *ayy, double *axy) {
  const double r1_3 = 1. / 3., r1_6 = 1. / 6., r1_12 = 1. / 12.;
  double xx = x1 * x2 + x2 * x3 + x1 * x3;
  double yy = y1 * y2 + y2 * y3 + y1 * y3;
  double xy1 = x1 * y1, xy2 = x2 * y2, xy3 = x3 * y3;
  double xy = xy1 + xy2 + xy3;
  *a = .5 * ((x2 - x1) * (y3 - y1) - (y2 - y1) * (x3 - x1)) * ac;
  *ax = r1_3 * (x1 + x2 + x3);
  *ay = r1_3 * (y1 + y2 + y3);
  *axx = r1_6 * (9. * *ax * *ax - xx);
  *ayy = r1_6 * (9. * *ay * *ay - yy);
  *axy = r1_12 * (9. * *ax * *ay + xy);
  *ax = *ax * *a;
  *ay = *ay * *a;
  *axx = *axx * *a;
  *ayy = *ayy * *a;
  *axy = *axy * *a;
}

/* penint, phy/mod_remap.F90:104-199 */
static void penint(double ac, double x1, double y1, double x2, double y2, double x3, double y3, double x4,
                   double y4, double x5, double y5, double *a, double *ax, double *ay, double *axx,
                   double *ayy, double *axy) {
  const double r1_3 = 1. / 3., r1_6 = 1. / 6., r1_12 = 1. / 12.;
  double xx123 = x1 * x2 + x2 * x3 + x1 * x3, yy123 = y1 * y2 + y2 * y3 + y1 * y3;
  double xx135 = x1 * x3 + x3 * x5 + x1 * x5, yy135 = y1 * y3 + y3 * y5 + y1 * y5;
  double xx345 = x3 * x4 + x4 * x5 + x3 * x5, yy345 = y3 * y4 + y4 * y5 + y3 * y5;
  double xy1 = x1 * y1, xy2 = x2 * y2, xy3 = x3 * y3, xy4 = x4 * y4, xy5 = x5 * y5;
  double xy123 = xy1 + xy2 + xy3, xy135 = xy1 + xy3 + xy5, xy345 = xy3 + xy4 + xy5;
  double a123 = .5 * ((x2 - x1) * (y3 - y1) - (y2 - y1) * (x3 - x1)) * ac;
  double a135 = .5 * ((x3 - x1) * (y5 - y1) - (y3 - y1) * (x5 - x1)) * ac;
  double a345 = .5 * ((x4 - x3) * (y5 - y3) - (y4 - y3) * (x5 - x3)) * ac;
  double ax123 = r1_3 * (x1 + x2 + x3), ay123 = r1_3 * (y1 + y2 + y3);
  double ax135 = r1_3 * (x1 + x3 + x5), ay135 = r1_3 * (y1 + y3 + y5);
  double ax345 = r1_3 * (x3 + x4 + x5), ay345 = r1_3 * (y3 + y4 + y5);
  double axx123 = r1_6 * (9. * ax123 * ax123 - xx123), ayy123 = r1_6 * (9. * ay123 * ay123 - yy123);
  double axy123 = r1_12 * (9. * ax123 * ay123 + xy123);
  double axx135 = r1_6 * (9. * ax135 * ax135 - xx135), ayy135 = r1_6 * (9. * ay135 * ay135 - yy135);
  double axy135 = r1_12 * (9. * ax135 * ay135 + xy135);
  double axx345 = r1_6 * (9. * ax345 * ax345 - xx345), ayy345 = r1_6 * (9. * ay345 * ay345 - yy345);
  double axy345 = r1_12 * (9. * ax345 * ay345 + xy345);
  *a = a123 + a135 + a345;
  *ax = ax123 * a123 + ax135 * a135 + ax345 * a345;
  *ay = ay123 * a123 + ay135 * a135 + ay345 * a345;
  *axx = axx123 * a123 + axx135 * a135 + axx345 * a345;
  *ayy = ayy123 * a123 + ayy135 * a135 + ayy345 * a345;
  *axy = axy123 * a123 + axy135 * a135 + axy345 * a345;
}

typedef struct {
  double *pup, *dx, *dy, *xd, *yd, *tx, *ty, *td, *sx, *sy, *sd, *cu, *cv, *cuc, *cvc, *fdu, *fdv, *ftu,
      *ftv, *fsu, *fsv;
  double *trx[MAXTR], *try_[MAXTR], *trd[MAXTR], *ftru[MAXTR], *ftrv[MAXTR];
} RemapWork;

#define L2(w, i, j) (w)[IX(S, i, j)]

/* neighbour indices restricted to wet points, phy/mod_remap.F90:365-376 (== mod_advect.F90:103-114) */
typedef struct { int iw, ie, js, jn, isw, jsw, ise, jse, inw, jnw, ine, jne; } Nbr;
static Nbr wet_nbr(const OState *S, int i, int j) {
  Nbr b;
  b.iw = i - A2(S, iu, i, j);
  b.ie = i + A2(S, iu, i + 1, j);
  b.js = j - A2(S, iv, i, j);
  b.jn = j + A2(S, iv, i, j + 1);
  b.isw = i * (1 - A2(S, ip, b.iw, b.js)) + b.iw * A2(S, ip, b.iw, b.js);
  b.jsw = j * (1 - A2(S, ip, b.iw, b.js)) + b.js * A2(S, ip, b.iw, b.js);
  b.ise = i * (1 - A2(S, ip, b.ie, b.js)) + b.ie * A2(S, ip, b.ie, b.js);
  b.jse = j * (1 - A2(S, ip, b.ie, b.js)) + b.js * A2(S, ip, b.ie, b.js);
  b.inw = i * (1 - A2(S, ip, b.iw, b.jn)) + b.iw * A2(S, ip, b.iw, b.jn);
  b.jnw = j * (1 - A2(S, ip, b.iw, b.jn)) + b.jn * A2(S, ip, b.iw, b.jn);
  b.ine = i * (1 - A2(S, ip, b.ie, b.jn)) + b.ie * A2(S, ip, b.ie, b.jn);
  b.jne = j * (1 - A2(S, ip, b.ie, b.jn)) + b.jn * A2(S, ip, b.ie, b.jn);
  return b;
}

static double max8(double a, double b, double c, double d, double e, double f, double g, double h) {
  return fmax2(fmax2(fmax2(fmax2(fmax2(fmax2(fmax2(a, b), c), d), e), f), g), h);
}
static double min8(double a, double b, double c, double d, double e, double f, double g, double h) {
  return fmin2(fmin2(fmin2(fmin2(fmin2(fmin2(fmin2(a, b), c), d), e), f), g), h);
}

/* limited gradient of a scalar field f (temp, saln, trc): phy/mod_remap.F90:412-439 */
static void limited_gradient(const OState *S, const double *f, const Nbr *b, int i, int j, double dxi,
                             double dyi, double xd, double yd, double *gx, double *gy, double *gd) {
#define F(ii_, jj_) f[IX(S, ii_, jj_)]
  double tx = (F(b->ie, j) - F(b->iw, j)) * dxi;
  double ty = (F(i, b->jn) - F(i, b->js)) * dyi;
  double q1 = tx * (-.5 - xd), q2 = tx * (.5 - xd), q3 = ty * (-.5 - yd), q4 = ty * (.5 - yd);
  double tgmx = fmax2(q1, q2) + fmax2(q3, q4);
  double tgmn = fmin2(q1, q2) + fmin2(q3, q4);
  double tfmx = fmax2(0., max8(F(b->isw, b->jsw), F(i, b->js), F(b->ise, b->jse), F(b->iw, j), F(b->ie, j),
                               F(b->inw, b->jnw), F(i, b->jn), F(b->ine, b->jne)) - F(i, j));
  double tfmn = fmin2(0., min8(F(b->isw, b->jsw), F(i, b->js), F(b->ise, b->jse), F(b->iw, j), F(b->ie, j),
                               F(b->inw, b->jnw), F(i, b->jn), F(b->ine, b->jne)) - F(i, j));
  if (tfmx > 0. && tfmn < 0.) {
    double q = fmin2(tfmx / fmax2(tfmx, tgmx), tfmn / fmin2(tfmn, tgmn));
    tx = tx * q;
    ty = ty * q;
    *gd = F(i, j) - tx * xd - ty * yd;
  } else {
    tx = 0.;
    ty = 0.;
    *gd = F(i, j);
  }
  *gx = tx;
  *gy = ty;
#undef F
}

/* one polygon's contribution from donor cell (ic,jc): phy/mod_remap.F90:700-742 and siblings */
static void add_contrib(const OState *S, const RemapWork *W, const double *dp, int ic, int jc, double pbface,
                        double a, double ax, double ay, double axx, double ayy, double axy, double *fd_acc,
                        double *ft_acc, double *fs_acc, double *ftr_acc) {
  double dl = fmin2(dp[IX(S, ic, jc)], fmax2(0., pbface - L2(W->pup, ic, jc)));
  double fd = a * dl + ax * L2(W->dx, ic, jc) + ay * L2(W->dy, ic, jc);
  *fd_acc = *fd_acc + fd;
  double qx = ax * dl + axx * L2(W->dx, ic, jc) + axy * L2(W->dy, ic, jc);
  double qy = ay * dl + axy * L2(W->dx, ic, jc) + ayy * L2(W->dy, ic, jc);
  *ft_acc = *ft_acc + fd * L2(W->td, ic, jc) + qx * L2(W->tx, ic, jc) + qy * L2(W->ty, ic, jc);
  *fs_acc = *fs_acc + fd * L2(W->sd, ic, jc) + qx * L2(W->sx, ic, jc) + qy * L2(W->sy, ic, jc);
  for (int nt = 0; nt < S->ntr; nt++)
    ftr_acc[nt] = ftr_acc[nt] + fd * L2(W->trd[nt], ic, jc) + qx * L2(W->trx[nt], ic, jc) +
                  qy * L2(W->try_[nt], ic, jc);
}

/* remap, phy/mod_remap.F90:205-1522 (mrg = 1).  k = layer index into trc (k+nn of the caller). */
static void remap(OState *S, RemapWork *W, const double *pbmin, const double *pbu, const double *pbv,
                  const double *plo, const double *cau, const double *cav, int mrg, double *dp, double *temp,
                  double *saln, double *uflx, double *vflx, double *utflx, double *vtflx, double *usflx,
                  double *vsflx, int k) {
  const int ii = S->ii, jj = S->jj, ntr = S->ntr;
  const double *scp2 = S->scp2, *scp2i = S->scp2i;
  double *trck[MAXTR];
  for (int nt = 0; nt < ntr; nt++) trck[nt] = &TRC(S, 1 - NBDY, 1 - NBDY, k, nt + 1);

  /* :297-337 */
  for (int j = 1 - mrg - 2; j <= jj + mrg + 2; j++) {
    for (int i = 1 - mrg - 2; i <= ii + mrg + 2; i++)
      if (A2(S, ip, i, j)) {
        dp[IX(S, i, j)] = fmax2(0., dp[IX(S, i, j)]) + DPEPS;
        L2(W->pup, i, j) = plo[IX(S, i, j)] - dp[IX(S, i, j)];
      }
    for (int i = 1 - mrg - 1; i <= ii + mrg + 1; i++) {
      L2(W->fdu, i, j) = 0.; L2(W->fdv, i, j) = 0.; L2(W->ftu, i, j) = 0.; L2(W->ftv, i, j) = 0.;
      L2(W->fsu, i, j) = 0.; L2(W->fsv, i, j) = 0.;
      for (int nt = 0; nt < ntr; nt++) { L2(W->ftru[nt], i, j) = 0.; L2(W->ftrv[nt], i, j) = 0.; }
      L2(W->cu, i, j) = 0.; L2(W->cv, i, j) = 0.;    /* only under use_TRC, :333-334 */
    }
  }

  /* :358-584 limited gradients, centre of mass */
  for (int j = 1 - mrg - 1; j <= jj + mrg + 1; j++)
    for (int i = 1 - mrg - 1; i <= ii + mrg + 1; i++) {
      if (!A2(S, ip, i, j)) continue;
      Nbr b = wet_nbr(S, i, j);
      double dxi = 1. / imax2(1, b.ie - b.iw);
      double dyi = 1. / imax2(1, b.jn - b.js);
      double pm = pbmin[IX(S, i, j)];
#define LIM(ii_, jj_) fmax2(DPEPS, fmin2(pm - L2(W->pup, ii_, jj_), dp[IX(S, ii_, jj_)]))
      double dpsw = LIM(b.isw, b.jsw), dps = LIM(i, b.js), dpse = LIM(b.ise, b.jse), dpw = LIM(b.iw, j);
      double dpc = LIM(i, j), dpe = LIM(b.ie, j), dpnw = LIM(b.inw, b.jnw), dpn = LIM(i, b.jn);
      double dpne = LIM(b.ine, b.jne);
#undef LIM
      double dx = (dpe - dpw) * dxi, dy = (dpn - dps) * dyi;
      double dgmx = .5 * (fabs(dx) + fabs(dy));
      double dfmx = fmax2(0., max8(dpsw, dps, dpse, dpw, dpe, dpnw, dpn, dpne) - dpc);
      double dfmn = fmin2(0., min8(dpsw, dps, dpse, dpw, dpe, dpnw, dpn, dpne) - dpc);
      double xd, yd;
      if (dfmx > 0. && dfmn < 0.) {
        double q = fmin2(dfmx / fmax2(dfmx, dgmx), dfmn / fmin2(dfmn, -dgmx));
        dx = dx * q;
        dy = dy * q;
        xd = dx / (12. * dp[IX(S, i, j)]);
        yd = dy / (12. * dp[IX(S, i, j)]);
      } else {
        dx = 0.; dy = 0.; xd = 0.; yd = 0.;
      }
      L2(W->dx, i, j) = dx; L2(W->dy, i, j) = dy; L2(W->xd, i, j) = xd; L2(W->yd, i, j) = yd;
      limited_gradient(S, temp, &b, i, j, dxi, dyi, xd, yd, &L2(W->tx, i, j), &L2(W->ty, i, j), &L2(W->td, i, j));
      limited_gradient(S, saln, &b, i, j, dxi, dyi, xd, yd, &L2(W->sx, i, j), &L2(W->sy, i, j), &L2(W->sd, i, j));
      for (int nt = 0; nt < ntr; nt++)
        limited_gradient(S, trck[nt], &b, i, j, dxi, dyi, xd, yd, &L2(W->trx[nt], i, j), &L2(W->try_[nt], i, j),
                         &L2(W->trd[nt], i, j));
    }

  /* :588-610 non-dimensional velocities */
  for (int j = 1 - mrg - 1; j <= jj + mrg + 1; j++)
    for (int i = 1 - mrg; i <= ii + mrg + 1; i++)
      if (A2(S, iu, i, j)) {
        if (cau[IX(S, i, j)] > 0.) L2(W->cu, i, j) = cau[IX(S, i, j)] * scp2i[IX(S, i - 1, j)];
        else L2(W->cu, i, j) = cau[IX(S, i, j)] * scp2i[IX(S, i, j)];
      }
  for (int j = 1 - mrg; j <= jj + mrg + 1; j++)
    for (int i = 1 - mrg - 1; i <= ii + mrg + 1; i++)
      if (A2(S, iv, i, j)) {
        if (cav[IX(S, i, j)] > 0.) L2(W->cv, i, j) = cav[IX(S, i, j)] * scp2i[IX(S, i, j - 1)];
        else L2(W->cv, i, j) = cav[IX(S, i, j)] * scp2i[IX(S, i, j)];
      }

  /* :623-659 corner velocities */
  for (int j = 1 - mrg; j <= jj + mrg + 1; j++)
    for (int i = 1 - mrg; i <= ii + mrg + 1; i++) {
      int nw = A2(S, ip, i - 1, j - 1) + A2(S, ip, i, j - 1) + A2(S, ip, i - 1, j) + A2(S, ip, i, j);
      double cuc, cvc;
      if (nw == 4) {
        if (L2(W->cu, i, j - 1) * L2(W->cu, i, j) <= 0.) cuc = 0.;
        else cuc = 2. * L2(W->cu, i, j - 1) * L2(W->cu, i, j) / (L2(W->cu, i, j - 1) + L2(W->cu, i, j));
        if (L2(W->cv, i - 1, j) * L2(W->cv, i, j) <= 0.) cvc = 0.;
        else cvc = 2. * L2(W->cv, i - 1, j) * L2(W->cv, i, j) / (L2(W->cv, i - 1, j) + L2(W->cv, i, j));
      } else if (nw == 2) {
        if (A2(S, ip, i - 1, j - 1) + A2(S, ip, i, j - 1) == 2) { cuc = L2(W->cu, i, j - 1); cvc = 0.; }
        else if (A2(S, ip, i - 1, j) + A2(S, ip, i, j) == 2) { cuc = L2(W->cu, i, j); cvc = 0.; }
        else if (A2(S, ip, i - 1, j - 1) + A2(S, ip, i - 1, j) == 2) { cuc = 0.; cvc = L2(W->cv, i - 1, j); }
        else if (A2(S, ip, i, j - 1) + A2(S, ip, i, j) == 2) { cuc = 0.; cvc = L2(W->cv, i, j); }
        else { cuc = 0.; cvc = 0.; }
      } else {
        cuc = 0.; cvc = 0.;
      }
      L2(W->cuc, i, j) = cuc;
      L2(W->cvc, i, j) = cvc;
    }

  double ftr[MAXTR];
  /* :667-1061 u-components of fluxes */
  for (int j = 1 - mrg; j <= jj + mrg; j++)
    for (int i = 1 - mrg; i <= ii + mrg + 1; i++) {
      if (!A2(S, iu, i, j)) continue;
      const double cu = L2(W->cu, i, j), cuc0 = L2(W->cuc, i, j), cuc1 = L2(W->cuc, i, j + 1);
      const double cvc0 = L2(W->cvc, i, j), cvc1 = L2(W->cvc, i, j + 1);
      double ym = -.5 * (cvc0 + cvc1);
      double xm = ((ym + .5) * cuc0 - (ym - .5) * cuc1 - 2. * cu) / (1. + cvc0 - cvc1);
      const int ic = cu > 0. ? i - 1 : i;
      const double sh = cu > 0. ? .5 : -.5;
      double fd = L2(W->fdu, i, j), ft = L2(W->ftu, i, j), fs = L2(W->fsu, i, j);
      for (int nt = 0; nt < ntr; nt++) ftr[nt] = L2(W->ftru[nt], i, j);
      double a, ax, ay, axx, ayy, axy, x2, y2, x4, y4;
      const double pb = pbu[IX(S, i, j)];
      if (cvc0 > 0.) {
        double xc0 = (xm * cvc0 - cuc0 * (ym + .5)) / (cvc0 + ym + .5);
        double xc1 = xc0 * scp2[IX(S, ic, j)] * scp2i[IX(S, ic, j - 1)];
        x4 = xc0 + sh;
        y4 = -.5;
        triint(scp2[IX(S, ic, j - 1)], xc1 + sh, .5, -cuc0 + sh, -cvc0 + .5, sh, .5, &a, &ax, &ay, &axx, &ayy, &axy);
        add_contrib(S, W, dp, ic, j - 1, pb, a, ax, ay, axx, ayy, axy, &fd, &ft, &fs, ftr);
      } else {
        x4 = -cuc0 + sh;
        y4 = -cvc0 - .5;
      }
      if (cvc1 < 0.) {
        double xc0 = (xm * cvc1 - cuc1 * (ym - .5)) / (cvc1 + ym - .5);
        double xc1 = xc0 * scp2[IX(S, ic, j)] * scp2i[IX(S, ic, j + 1)];
        x2 = xc0 + sh;
        y2 = .5;
        triint(scp2[IX(S, ic, j + 1)], xc1 + sh, -.5, sh, -.5, -cuc1 + sh, -cvc1 - .5, &a, &ax, &ay, &axx, &ayy, &axy);
        add_contrib(S, W, dp, ic, j + 1, pb, a, ax, ay, axx, ayy, axy, &fd, &ft, &fs, ftr);
      } else {
        x2 = -cuc1 + sh;
        y2 = -cvc1 + .5;
      }
      penint(scp2[IX(S, ic, j)], sh, .5, x2, y2, xm + sh, ym, x4, y4, sh, -.5, &a, &ax, &ay, &axx, &ayy, &axy);
      add_contrib(S, W, dp, ic, j, pb, a, ax, ay, axx, ayy, axy, &fd, &ft, &fs, ftr);
      L2(W->fdu, i, j) = fd; L2(W->ftu, i, j) = ft; L2(W->fsu, i, j) = fs;
      for (int nt = 0; nt < ntr; nt++) L2(W->ftru[nt], i, j) = ftr[nt];
      /* :1054-1056 */
      uflx[IX(S, i, j)] = uflx[IX(S, i, j)] + fd;
      utflx[IX(S, i, j)] = utflx[IX(S, i, j)] + ft;
      usflx[IX(S, i, j)] = usflx[IX(S, i, j)] + fs;
    }

  /* :1065-1462 v-components of fluxes */
  for (int j = 1 - mrg; j <= jj + mrg + 1; j++)
    for (int i = 1 - mrg; i <= ii + mrg; i++) {
      if (!A2(S, iv, i, j)) continue;
      const double cv = L2(W->cv, i, j), cuc0 = L2(W->cuc, i, j), cuc1 = L2(W->cuc, i + 1, j);
      const double cvc0 = L2(W->cvc, i, j), cvc1 = L2(W->cvc, i + 1, j);
      double xm = -.5 * (cuc0 + cuc1);
      double ym = ((xm + .5) * cvc0 - (xm - .5) * cvc1 - 2. * cv) / (1. + cuc0 - cuc1);
      const int jc = cv > 0 ? j - 1 : j;
      const double sh = cv > 0 ? .5 : -.5;
      double fd = L2(W->fdv, i, j), ft = L2(W->ftv, i, j), fs = L2(W->fsv, i, j);
      for (int nt = 0; nt < ntr; nt++) ftr[nt] = L2(W->ftrv[nt], i, j);
      double a, ax, ay, axx, ayy, axy, x2, y2, x4, y4;
      const double pb = pbv[IX(S, i, j)];
      if (cuc0 > 0.) {
        double yc0 = (ym * cuc0 - cvc0 * (xm + .5)) / (cuc0 + xm + .5);
        double yc1 = yc0 * scp2[IX(S, i, jc)] * scp2i[IX(S, i - 1, jc)];
        x2 = -.5;
        y2 = yc0 + sh;
        triint(scp2[IX(S, i - 1, jc)], .5, yc1 + sh, .5, sh, -cuc0 + .5, -cvc0 + sh, &a, &ax, &ay, &axx, &ayy, &axy);
        add_contrib(S, W, dp, i - 1, jc, pb, a, ax, ay, axx, ayy, axy, &fd, &ft, &fs, ftr);
      } else {
        x2 = -cuc0 - .5;
        y2 = -cvc0 + sh;
      }
      if (cuc1 < 0.) {
        double yc0 = (ym * cuc1 - cvc1 * (xm - .5)) / (cuc1 + xm - .5);
        double yc1 = yc0 * scp2[IX(S, i, jc)] * scp2i[IX(S, i + 1, jc)];
        x4 = .5;
        y4 = yc0 + sh;
        triint(scp2[IX(S, i + 1, jc)], -.5, yc1 + sh, -cuc1 - .5, -cvc1 + sh, -.5, sh, &a, &ax, &ay, &axx, &ayy, &axy);
        add_contrib(S, W, dp, i + 1, jc, pb, a, ax, ay, axx, ayy, axy, &fd, &ft, &fs, ftr);
      } else {
        x4 = -cuc1 + .5;
        y4 = -cvc1 + sh;
      }
      penint(scp2[IX(S, i, jc)], -.5, sh, x2, y2, xm, ym + sh, x4, y4, .5, sh, &a, &ax, &ay, &axx, &ayy, &axy);
      add_contrib(S, W, dp, i, jc, pb, a, ax, ay, axx, ayy, axy, &fd, &ft, &fs, ftr);
      L2(W->fdv, i, j) = fd; L2(W->ftv, i, j) = ft; L2(W->fsv, i, j) = fs;
      for (int nt = 0; nt < ntr; nt++) L2(W->ftrv[nt], i, j) = ftr[nt];
      /* :1455-1457 -- note: assignment, not accumulation, for the v-components */
      vflx[IX(S, i, j)] = fd;
      vtflx[IX(S, i, j)] = ft;
      vsflx[IX(S, i, j)] = fs;
    }

  /* :1468-1520 update fields */
  for (int j = 1 - mrg; j <= jj + mrg; j++)
    for (int i = 1 - mrg; i <= ii + mrg; i++) {
      if (!A2(S, ip, i, j)) continue;
      const size_t c = IX(S, i, j), e = IX(S, i + 1, j), nb = IX(S, i, j + 1);
      double q = dp[c];
      dp[c] = q - (W->fdu[e] - W->fdu[c] + W->fdv[nb] - W->fdv[c]) * scp2i[c];
      temp[c] = (q * temp[c] - (W->ftu[e] - W->ftu[c] + W->ftv[nb] - W->ftv[c]) * scp2i[c]) / dp[c];
      saln[c] = (q * saln[c] - (W->fsu[e] - W->fsu[c] + W->fsv[nb] - W->fsv[c]) * scp2i[c]) / dp[c];
      for (int nt = 0; nt < ntr; nt++)
        if (!orc_skip_adv(S, nt + 1)) trck[nt][c] = (q * trck[nt][c] -
                       (W->ftru[nt][e] - W->ftru[nt][c] + W->ftrv[nt][nb] - W->ftrv[nt][c]) * scp2i[c]) / dp[c];
      dp[c] = fmax2(0., dp[c] - DPEPS);
    }
}

/* advect, phy/mod_advect.F90:59-189 */
void orc_advect(OState *S, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)k1m;
  const int ii = S->ii, jj = S->jj, kk = S->kk, ntr = S->ntr;
  const size_t lev = (size_t)S->nplane;
  const double delt1 = S->delt1, dlt = S->dlt;
  /* :71-94 flux areas */
  for (int j = 1; j <= jj; j++)
    for (int k = 1; k <= kk; k++) {
      int km = k + mm, kn = k + nn;
      for (int i = 1; i <= ii; i++) {
        if (A2(S, iu, i, j)) {
          double dtdl = delt1 * A2(S, scuy, i, j);
          double ca_tmp = A3(S, u, i, j, km) * dtdl + A3(S, ubflxs_p, i, j, m) * dlt / A3(S, pbu, i, j, m) +
                          (A3(S, umfltd, i, j, km) + A3(S, umflsm, i, j, km)) / fmax2(ONEMM, A3(S, dpu, i, j, kn));
          A3(S, cau, i, j, k) = fmax2(-A2(S, umax, i, j) * dtdl, fmin2(A2(S, umax, i, j) * dtdl, ca_tmp));
        }
        if (A2(S, iv, i, j)) {
          double dtdl = delt1 * A2(S, scvx, i, j);
          double ca_tmp = A3(S, v, i, j, km) * dtdl + A3(S, vbflxs_p, i, j, m) * dlt / A3(S, pbv, i, j, m) +
                          (A3(S, vmfltd, i, j, km) + A3(S, vmflsm, i, j, km)) / fmax2(ONEMM, A3(S, dpv, i, j, kn));
          A3(S, cav, i, j, k) = fmax2(-A2(S, vmax, i, j) * dtdl, fmin2(A2(S, vmax, i, j) * dtdl, ca_tmp));
        }
      }
    }
  if (S->advmth != 0) { abort(); }
  /* :100-121 pbmin */
  double *pbmin = (double *)calloc(lev, sizeof(double));
  const double *pbot = S->p + lev * kk;
  for (int j = -1; j <= jj + 2; j++)
    for (int i = -1; i <= ii + 2; i++) {
      if (!A2(S, ip, i, j)) continue;
      Nbr b = wet_nbr(S, i, j);
#define PB(ii_, jj_) pbot[IX(S, ii_, jj_)]
      pbmin[IX(S, i, j)] = fmin2(fmin2(fmin2(fmin2(fmin2(fmin2(fmin2(fmin2(
          PB(b.isw, b.jsw), PB(i, b.js)), PB(b.ise, b.jse)), PB(b.iw, j)), PB(i, j)), PB(b.ie, j)),
          PB(b.inw, b.jnw)), PB(i, b.jn)), PB(b.ine, b.jne));
#undef PB
    }
  orc_xctilr(S, S->cau, 1, kk, 3, 3, 13);                              /* :124-131 */
  orc_xctilr(S, S->cav, 1, kk, 3, 3, 14);
  for (int nt = 1; nt <= ntr; nt++)
    if (!orc_skip_adv(S, nt)) orc_xctilr(S, S->trc + lev * ((size_t)(k1n - 1) + 2 * kk * (nt - 1)), 1, kk, 3, 3, 1);
  RemapWork W;
  double **all = (double **)&W;
  const int nw = sizeof(RemapWork) / sizeof(double *);
  for (int x = 0; x < nw; x++) all[x] = (double *)calloc(lev, sizeof(double));
  for (int k = 1; k <= kk; k++) {                                      /* :135-152 */
    int km = k + mm, kn = k + nn;
    remap(S, &W, pbmin, S->pbu + lev * (n - 1), S->pbv + lev * (n - 1), S->p + lev * k, S->cau + lev * (k - 1),
          S->cav + lev * (k - 1), 1, S->dp + lev * (kn - 1), S->temp + lev * (kn - 1), S->saln + lev * (kn - 1),
          S->uflx + lev * (km - 1), S->vflx + lev * (km - 1), S->utflx + lev * (km - 1),
          S->vtflx + lev * (km - 1), S->usflx + lev * (km - 1), S->vsflx + lev * (km - 1), kn);
  }
  for (int x = 0; x < nw; x++) free(all[x]);
  free(pbmin);
}
