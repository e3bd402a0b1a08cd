// difest_isobml, phy/mod_difest.F90:735-809 -- the part in front of the diffusivity estimates:
//   halo updates and the interface pressure (:750-772; blomgpu_halo_difest),
//   ustar3 = ustar**3 (:778-786),
//   niw_ke_tendency, phy/mod_niw.F90:121-217: energy input by near-inertial waves from the change of the mixed layer's
//   kinetic energy over a step, of the velocity deviations from a running time average (averaging time: the inertial period
//   times 2, the period limited to the one at 10 N/S) -> idkedt, which mxlayr's turbulent kinetic energy balance reads.
// NOT built: difest_common_iso, difest_vertical_iso, difest_lateral_iso (:353-586, :2040-3084: Richardson numbers, the
// Eden-Greatbatch lateral and the shear / tidal / background vertical diffusivities): difint, difiso, difdia, difwgt stay as
// uploaded.  Their module needs the CVMix library at module level, so nothing of them can be cross-checked in this image.
// Roofline: HBM; ~16 two-dimensional planes + 8 three-dimensional levels.
#include "blomgpu_internal.h"
#include "../../include/blomgpu.h"

#define PLANE_IJ(V)                                                        \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_;                                                     \
  (void)i; (void)j; (void)c

#define GRAV 9.806
#define ALPHA0 1.e-3
#define PI_BLOM 3.1415926536      // phy/mod_constants.F90:39

__global__ void k_difest_ustar3(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const double us = V.f[F_ustar][c];
  V.f[F_ustar3][c] = us * us * us;
}

// phy/mod_niw.F90:136-205: u-points i = 1..ii+1 -> util1, v-points j = 1..jj+1 -> util2
__global__ void k_niw_uv(const DevView *__restrict__ Vp, int m, int mm) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  const size_t np = V.nplane, om = (size_t)(m - 1) * np;
  const int mmm = (m - 1) * 2;
  const double delt1 = V.P.delt1, dlt = V.P.dlt, ipfac = 2., cori10 = 2.5256e-5;
  gcd_t coriop = V.f[F_coriop];
  if (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii + 1 && V.m[I_iu][c]) {
    const double ubt = V.f[F_ubflxs_p][c + om] * dlt / (delt1 * V.f[F_scuy][c] * V.f[F_pbu][c + om]);
    const double uml1t = V.f[F_u][c + (size_t)mm * np] + ubt;
    const double uml2t = V.f[F_u][c + (size_t)(1 + mm) * np] + ubt;
    const double q = delt1 * fmax2(cori10, fabs(.5 * (coriop[c - 1] + coriop[c]))) / (ipfac * 2. * PI_BLOM);
    gd_t res = V.f[F_umlres] + c, ml = V.f[F_uml] + c;
    double r1 = res[0] + uml1t;
    const double uml1a = r1 * q;
    res[0] = r1 * (1. - q);
    double r2 = res[np] + uml2t;
    const double uml2a = r2 * q;
    res[np] = r2 * (1. - q);
    const double o1 = ml[(size_t)mmm * np], o2 = ml[(size_t)(1 + mmm) * np];
    V.f[F_util1][c] = ((uml1t - uml1a) * (uml1t - uml1a) - (o1 - uml1a) * (o1 - uml1a)) * V.f[F_dpu][c + (size_t)mm * np] +
                      ((uml2t - uml2a) * (uml2t - uml2a) - (o2 - uml2a) * (o2 - uml2a)) * V.f[F_dpu][c + (size_t)(1 + mm) * np];
    ml[(size_t)mmm * np] = uml1t;
    ml[(size_t)(1 + mmm) * np] = uml2t;
  }
  if (j >= 1 && j <= V.jj + 1 && i >= 1 && i <= V.ii && V.m[I_iv][c]) {
    const double vbt = V.f[F_vbflxs_p][c + om] * dlt / (delt1 * V.f[F_scvx][c] * V.f[F_pbv][c + om]);
    const double vml1t = V.f[F_v][c + (size_t)mm * np] + vbt;
    const double vml2t = V.f[F_v][c + (size_t)(1 + mm) * np] + vbt;
    const double q = delt1 * fmax2(cori10, fabs(.5 * (coriop[c - V.ni] + coriop[c]))) / (ipfac * 2. * PI_BLOM);
    gd_t res = V.f[F_vmlres] + c, ml = V.f[F_vml] + c;
    double r1 = res[0] + vml1t;
    const double vml1a = r1 * q;
    res[0] = r1 * (1. - q);
    double r2 = res[np] + vml2t;
    const double vml2a = r2 * q;
    res[np] = r2 * (1. - q);
    const double o1 = ml[(size_t)mmm * np], o2 = ml[(size_t)(1 + mmm) * np];
    V.f[F_util2][c] = ((vml1t - vml1a) * (vml1t - vml1a) - (o1 - vml1a) * (o1 - vml1a)) * V.f[F_dpv][c + (size_t)mm * np] +
                      ((vml2t - vml2a) * (vml2t - vml2a) - (o2 - vml2a) * (o2 - vml2a)) * V.f[F_dpv][c + (size_t)(1 + mm) * np];
    ml[(size_t)mmm * np] = vml1t;
    ml[(size_t)(1 + mmm) * np] = vml2t;
  }
}

// :207-217
__global__ void k_niw_idkedt(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  gci_t iu = V.m[I_iu], iv = V.m[I_iv];
  gcd_t u1 = V.f[F_util1], u2 = V.f[F_util2];
  const int ni = V.ni;
  const int su = iu[c] + iu[c + 1], sv = iv[c] + iv[c + ni];
  V.f[F_idkedt][c] = fabs((u1[c] * iu[c] + u1[c + 1] * iu[c + 1]) / (su > 1 ? su : 1) + (u2[c] * iv[c] + u2[c + ni] * iv[c + ni]) / (sv > 1 ? sv : 1)) *
                     ALPHA0 / (2. * GRAV * V.P.delt1);
}

int st_niw_ke_tendency(blomgpu_ctx *c, int m, int mm) {
  const DevView &h = c->h;
  hipLaunchKernelGGL(k_niw_uv, plane_grid(h), dim3(256), 0, c->stream, c->d, m, mm);
  hipLaunchKernelGGL(k_niw_idkedt, plane_grid(h), dim3(256), 0, c->stream, c->d);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// the part of difest_isobml that is built: see the header
int st_difest_isobml_pre(blomgpu_ctx *c, int m, int n, int mm, int nn) {
  (void)n;
  if (int rc = blomgpu_halo_difest(c, nn)) return rc;
  TimeScope ts(c, "difest");
  hipLaunchKernelGGL(k_difest_ustar3, plane_grid(c->h), dim3(256), 0, c->stream, c->d);
  return st_niw_ke_tendency(c, m, mm);
}
