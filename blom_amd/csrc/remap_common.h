// Pieces of the incremental remapping shared by its two forms (stage_advect.hip: one kernel per sweep;
// stage_remap_tile.hip: gradient and flux sweep in one LDS-tiled kernel).  phy/mod_remap.F90:40-199.
#pragma once
#include "blomgpu_internal.h"

#define DPEPS 1.e-12   // phy/mod_remap.F90:40
#define ONEMM 9.806
#define MAXTR 4   // tracer accumulators live in registers: every loop over them has a
                  // compile-time trip count (a runtime-indexed array would go to scratch)

// work-space slots (fields of kk levels)
#define G_DX 0
#define G_DY 1
#define G_TX 2
#define G_TY 3
#define G_TD 4
#define G_SX 5
#define G_SY 6
#define G_SD 7
#define G_TRX(nt) (8 + 3 * (nt))
#define G_TRY(nt) (9 + 3 * (nt))
#define G_TRD(nt) (10 + 3 * (nt))
#define F_BASE(ntr) (8 + 3 * (ntr))
#define W_FDU(ntr) (F_BASE(ntr) + 0)
#define W_FDV(ntr) (F_BASE(ntr) + 1)
#define W_FTU(ntr) (F_BASE(ntr) + 2)
#define W_FTV(ntr) (F_BASE(ntr) + 3)
#define W_FSU(ntr) (F_BASE(ntr) + 4)
#define W_FSV(ntr) (F_BASE(ntr) + 5)
#define W_FTRU(ntr, nt) (F_BASE(ntr) + 6 + 2 * (nt))
#define W_FTRV(ntr, nt) (F_BASE(ntr) + 7 + 2 * (nt))
static_assert(F_BASE(3) == R_BASE(3) && F_BASE(0) == R_BASE(0), "R_DP.. (blomgpu_internal.h) reuse the flux planes' slots");

__device__ inline double max8(double a, double b, double c, double d, double e, double f, double g, double h) {
  return fmax2(fmax2(fmax2(fmax2(fmax2(fmax2(fmax2(a, b), c), d), e), f), g), h);
}
__device__ inline double min8(double a, double b, double c, double d, double e, double f, double g, double h) {
  return fmin2(fmin2(fmin2(fmin2(fmin2(fmin2(fmin2(a, b), c), d), e), f), g), h);
}


// triint, mod_remap.F90:53-102
__device__ inline void triint(double ac, double x1, double y1, double x2, double y2, double x3, double y3,
                              double &a, double &ax, double &ay, double &axx, double &ayy, double &axy) {
  const double r1_3 = 1. / 3., r1_6 = 1. / 6., r1_12 = 1. / 12.;
  const double xx = x1 * x2 + x2 * x3 + x1 * x3;
  const double yy = y1 * y2 + y2 * y3 + y1 * y3;
  const double xy1 = x1 * y1, xy2 = x2 * y2, xy3 = x3 * y3;
  const double xy = xy1 + xy2 + xy3;
  a = .5 * ((x2 - x1) * (y3 - y1) - (y2 - y1) * (x3 - x1)) * ac;
  ax = r1_3 * (x1 + x2 + x3);
  ay = r1_3 * (y1 + y2 + y3);
  axx = r1_6 * (9. * ax * ax - xx);
  ayy = r1_6 * (9. * ay * ay - yy);
  axy = r1_12 * (9. * ax * ay + xy);
  ax = ax * a;
  ay = ay * a;
  axx = axx * a;
  ayy = ayy * a;
  axy = axy * a;
}

// penint, mod_remap.F90:104-199
__device__ inline void penint(double ac, double x1, double y1, double x2, double y2, double x3, double y3,
                              double x4, double y4, double x5, double y5, double &a, double &ax, double &ay,
                              double &axx, double &ayy, double &axy) {
  const double r1_3 = 1. / 3., r1_6 = 1. / 6., r1_12 = 1. / 12.;
  const double xx123 = x1 * x2 + x2 * x3 + x1 * x3, yy123 = y1 * y2 + y2 * y3 + y1 * y3;
  const double xx135 = x1 * x3 + x3 * x5 + x1 * x5, yy135 = y1 * y3 + y3 * y5 + y1 * y5;
  const double xx345 = x3 * x4 + x4 * x5 + x3 * x5, yy345 = y3 * y4 + y4 * y5 + y3 * y5;
  const double xy1 = x1 * y1, xy2 = x2 * y2, xy3 = x3 * y3, xy4 = x4 * y4, xy5 = x5 * y5;
  const double xy123 = xy1 + xy2 + xy3, xy135 = xy1 + xy3 + xy5, xy345 = xy3 + xy4 + xy5;
  const double a123 = .5 * ((x2 - x1) * (y3 - y1) - (y2 - y1) * (x3 - x1)) * ac;
  const double a135 = .5 * ((x3 - x1) * (y5 - y1) - (y3 - y1) * (x5 - x1)) * ac;
  const double a345 = .5 * ((x4 - x3) * (y5 - y3) - (y4 - y3) * (x5 - x3)) * ac;
  const double ax123 = r1_3 * (x1 + x2 + x3), ay123 = r1_3 * (y1 + y2 + y3);
  const double ax135 = r1_3 * (x1 + x3 + x5), ay135 = r1_3 * (y1 + y3 + y5);
  const double ax345 = r1_3 * (x3 + x4 + x5), ay345 = r1_3 * (y3 + y4 + y5);
  const double axx123 = r1_6 * (9. * ax123 * ax123 - xx123), ayy123 = r1_6 * (9. * ay123 * ay123 - yy123);
  const double axy123 = r1_12 * (9. * ax123 * ay123 + xy123);
  const double axx135 = r1_6 * (9. * ax135 * ax135 - xx135), ayy135 = r1_6 * (9. * ay135 * ay135 - yy135);
  const double axy135 = r1_12 * (9. * ax135 * ay135 + xy135);
  const double axx345 = r1_6 * (9. * ax345 * ax345 - xx345), ayy345 = r1_6 * (9. * ay345 * ay345 - yy345);
  const double axy345 = r1_12 * (9. * ax345 * ay345 + xy345);
  a = a123 + a135 + a345;
  ax = ax123 * a123 + ax135 * a135 + ax345 * a345;
  ay = ay123 * a123 + ay135 * a135 + ay345 * a345;
  axx = axx123 * a123 + axx135 * a135 + axx345 * a345;
  ayy = ayy123 * a123 + ayy135 * a135 + ayy345 * a345;
  axy = axy123 * a123 + axy135 * a135 + axy345 * a345;
}

struct Acc {
  double fd, ft, fs, ftr[MAXTR];
};

