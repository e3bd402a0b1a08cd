// cppm -- compatible piecewise parabolic method, phy/mod_cppm.F90 (advmth = 'cppm', the default of
// the hybrid-coordinate configurations; the reference's own default variant: cppm_compatibility =
// 'full', cppm_limiting = 'non_oscillatory', :44-47).
//
// Strang-split 1-D transport of dp and dp*tracer (:2748-2834): an i-sweep and a j-sweep in an order
// that alternates with the step parity, the second sweep with a divergence correction of the
// thickness.  One sweep (cppm_fc_nosc_i :1470-1623 / _j :1625-1785) is, per row and layer,
//   h_edges_nosc            4th-order thickness edge values + non-oscillatory limiting   (:361-434)
//   parabola_coeffs_fc_nosc tracer edge values compatible with the thickness reconstruction (a 4x4
//                           LU per edge, 9 stencil cases), limiting, parabola coefficients (:490-818)
//   flux_integration        upstream parabola integrals over the swept area               (:1373-1468)
//   divergence update of dp, T, S, tracers; flux accumulation.
// Rows and layers are independent and every step is a short stencil along the sweep direction, so
// each becomes an (i,j,k)-parallel kernel, templated on the direction (stride 1 or ni); the
// reference's 1-D temporaries become work-space fields:
//   k_cppm_hm      hm = max(0,dp)+dpeps [/ (1 - divergence of the other direction's Courant number)]
//   k_cppm_hedges  hel_3d, her_3d                                  (halo update, 4 cells along the sweep)
//   k_cppm_tedge   tracer edge values te(nt) at every edge
//   k_cppm_parab   limiting and parabola coefficients hpc0-2, tpc0-2(nt)
//   k_cppm_flux    hf, htf(nt) at every edge; accumulation of uflx,utflx,usflx / vflx,..
//   k_cppm_update  dp, temp, saln, trc
// init_cppm (:2504-2746) -- stencil tags and the metric-dependent coefficient tables -- is
// k_cppm_init + halo updates; the j-tables keep (i,j) order (the reference's "_perm" layout) and
// are addressed with stride ni, which is what its final transposition (:2717-2733) achieves on the
// CPU.  Algorithmic bytes per step: (48 + 4 ntr) F.  Roofline: HBM.
// All four variants of the reference are instantiated from the same templates: limiting non_oscillatory
// (reach 4) / monotonic (reach 3) x compatibility full (LU edge coefficients, hel_3d/her_3d exchanged) /
// partial (thickness edge coefficients re-used for tracers).  With the arctic patch (nreg=2) the tags, edge coefficients
// and thickness edge values next to the seam change roles (:1531-1541, :1686-1704, :1848-1858, :2002-2020, :2650-2722).
#include "blomgpu_internal.h"

#define DPEPS 1.e-12

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

enum { ST_0000 = 0, ST_1111, ST_1110, ST_0111, ST_1100, ST_0110, ST_0011, ST_0100, ST_0010 };   // :60-68

// work-space slots
enum { W_HM = 0, W_HF, W_HPC0, W_HPC1, W_HPC2, W_TE, /* + ntl */ };
#define W_HTF(ntl) (W_TE + (ntl))
#define W_TPC0(ntl) (W_TE + 2 * (ntl))
#define W_TPC1(ntl) (W_TE + 3 * (ntl))
#define W_TPC2(ntl) (W_TE + 4 * (ntl))
#define W_NSLOT(ntl) (W_TE + 5 * (ntl))

// ---- init_cppm: set_stencil_coeffs (:101-320), set_slope_coeffs (:322-341), set_d2_mask (:343-359) ----
struct CppmTab {
  int *stencil;
  double *hevc[4], *ssc, *scc, *d2m, *tmc0, *tmcl, *tmcr;   // tmc*: 12 planes each
};

__device__ inline CppmTab cppm_tab(const DevView &V, int dir) {
  CppmTab T;
  T.stencil = dir ? V.m[I_cppm_stj] : V.m[I_cppm_sti];
  T.hevc[0] = dir ? V.f[F_hevc1j] : V.f[F_hevc1i];
  T.hevc[1] = dir ? V.f[F_hevc2j] : V.f[F_hevc2i];
  T.hevc[2] = dir ? V.f[F_hevc3j] : V.f[F_hevc3i];
  T.hevc[3] = dir ? V.f[F_hevc4j] : V.f[F_hevc4i];
  T.ssc = dir ? V.f[F_sscj] : V.f[F_ssci];
  T.scc = dir ? V.f[F_sccj] : V.f[F_scci];
  T.d2m = dir ? V.f[F_d2mj] : V.f[F_d2mi];
  T.tmc0 = dir ? V.f[F_tmc0j] : V.f[F_tmc0i];
  T.tmcl = dir ? V.f[F_tmclj] : V.f[F_tmcli];
  T.tmcr = dir ? V.f[F_tmcrj] : V.f[F_tmcri];
  return T;
}

__global__ void k_cppm_init(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  const int dir = by_;
  const size_t np = V.nplane;
  CppmTab T = cppm_tab(V, dir);
  // :2553-2574 everything zero / stencil_0000 outside the interior (halos come from xctilr afterwards)
  T.stencil[c] = ST_0000;
  for (int q = 0; q < 4; q++) T.hevc[q][c] = 0.;
  T.ssc[c] = 0.; T.scc[c] = 0.; T.d2m[c] = 0.;
  for (int q = 0; q < 12; q++) { T.tmc0[c + q * np] = 0.; T.tmcl[c + q * np] = 0.; T.tmcr[c + q * np] = 0.; }
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const int sd = dir ? V.ni : 1;
  gcd_t dxa = dir ? V.f[F_scpy] : V.f[F_scpx];
  gci_t ip = V.m[I_ip];
  const int m1 = ip[c - 2 * sd], m2 = ip[c - sd], m3 = ip[c], m4 = ip[c + sd];
  const double dx1 = dxa[c - 2 * sd], dx2 = dxa[c - sd], dx3 = dxa[c], dx4 = dxa[c + sd];
  const double c1_2 = 1. / 2., c2_3 = 2. / 3., c3_4 = 3. / 4., c1_4 = 1. / 4., c1_5 = 1. / 5., c1_6 = 1. / 6.,
               c1_10 = 1. / 10., c1_12 = 1. / 12., c1_15 = 1. / 15., c1_20 = 1. / 20.;
  double a12 = -dx2 - c1_2 * dx1;                                                     // :121-132
  double a22 = -c1_2 * dx2;
  double a32 = c1_2 * dx3;
  double a42 = dx3 + c1_2 * dx4;
  double a13 = a12 * a12 + c1_12 * dx1 * dx1;
  double a23 = -c2_3 * a22 * dx2;
  double a33 = c2_3 * a32 * dx3;
  double a43 = a42 * a42 + c1_12 * dx4 * dx4;
  double a14 = (a13 + c1_6 * dx1 * dx1) * a12;
  double a24 = -c3_4 * a23 * dx2;
  double a34 = c3_4 * a33 * dx3;
  double a44 = (a43 + c1_6 * dx4 * dx4) * a42;
  double tl[12], tr[12], t0[12];
  tl[0] = -c1_12 * dx1;                                                               // :137-148
  tl[1] = (c1_10 * dx1 + c1_6 * dx2) * dx1;
  tl[2] = -(c1_10 * (dx1 + 3. * dx2) * dx1 + c1_4 * (dx2 * dx2)) * dx1;
  tl[3] = -c1_12 * dx2;
  tl[4] = c1_10 * (dx2 * dx2);
  tl[5] = -c1_10 * (dx2 * dx2 * dx2);
  tl[6] = -c1_12 * dx3;
  tl[7] = -c1_15 * (dx3 * dx3);
  tl[8] = -c1_20 * (dx3 * dx3 * dx3);
  tl[9] = -c1_12 * dx4;
  tl[10] = -(c1_15 * dx4 + c1_6 * dx3) * dx4;
  tl[11] = -(c1_5 * (c1_4 * dx4 + dx3) * dx4 + c1_4 * (dx3 * dx3)) * dx4;
  tr[0] = c1_12 * dx1;                                                                // :150-161
  tr[1] = -(c1_15 * dx1 + c1_6 * dx2) * dx1;
  tr[2] = (c1_5 * (c1_4 * dx1 + dx2) * dx1 + c1_4 * (dx2 * dx2)) * dx1;
  tr[3] = c1_12 * dx2;
  tr[4] = -c1_15 * (dx2 * dx2);
  tr[5] = c1_20 * (dx2 * dx2 * dx2);
  tr[6] = c1_12 * dx3;
  tr[7] = c1_10 * (dx3 * dx3);
  tr[8] = c1_10 * (dx3 * dx3 * dx3);
  tr[9] = c1_12 * dx4;
  tr[10] = (c1_10 * dx4 + c1_6 * dx3) * dx4;
  tr[11] = (c1_10 * (dx4 + 3. * dx3) * dx4 + c1_4 * (dx3 * dx3)) * dx4;
  t0[0] = a12; t0[1] = a13 - tl[1] - tr[1]; t0[2] = a14 - tl[2] - tr[2];              // :163-174
  t0[3] = a22; t0[4] = a23 - tl[4] - tr[4]; t0[5] = a24 - tl[5] - tr[5];
  t0[6] = a32; t0[7] = a33 - tl[7] - tr[7]; t0[8] = a34 - tl[8] - tr[8];
  t0[9] = a42; t0[10] = a43 - tl[10] - tr[10]; t0[11] = a44 - tl[11] - tr[11];
#pragma unroll
  for (int q = 0; q < 12; q++) { T.tmc0[c + q * np] = t0[q]; T.tmcl[c + q * np] = tl[q]; T.tmcr[c + q * np] = tr[q]; }
  int st;
  double h1, h2, h3, h4;
  if (m1 == 1 && m2 == 1 && m3 == 1 && m4 == 1) {                                     // :176-203
    st = ST_1111;
    a22 = a22 - a12; a32 = a32 - a12; a42 = a42 - a12;
    a23 = (a23 - a13) / a22;
    a33 = a33 - a13 - a23 * a32;
    a43 = a43 - a13 - a23 * a42;
    a24 = (a24 - a14) / a22;
    a34 = a34 - a14 - a24 * a32;
    a44 = a44 - a14 - a24 * a42;
    a34 = a34 / a33;
    a44 = a44 - a34 * a43;
    h2 = -a12;
    h3 = -a13 - a23 * h2;
    h4 = -a14 - a24 * h2 - a34 * h3;
    h4 = h4 / a44;
    h3 = (h3 - a43 * h4) / a33;
    h2 = (h2 - a32 * h3 - a42 * h4) / a22;
    h1 = 1. - h2 - h3 - h4;
  } else if (m1 == 1 && m2 == 1 && m3 == 1 && m4 == 0) {                              // :205-225
    st = ST_1110;
    a22 = a22 - a12; a32 = a32 - a12;
    a23 = (a23 - a13) / a22;
    a33 = a33 - a13 - a23 * a32;
    h2 = -a12;
    h3 = -a13 - a23 * h2;
    h3 = h3 / a33;
    h2 = (h2 - a32 * h3) / a22;
    h1 = 1. - h2 - h3;
    h4 = 0.;
  } else if (m1 == 0 && m2 == 1 && m3 == 1 && m4 == 1) {                              // :227-247
    st = ST_0111;
    a32 = a32 - a22; a42 = a42 - a22;
    a33 = (a33 - a23) / a32;
    a43 = a43 - a23 - a33 * a42;
    h3 = -a22;
    h4 = -a23 - a33 * h3;
    h4 = h4 / a43;
    h3 = (h3 - a42 * h4) / a32;
    h2 = 1. - h3 - h4;
    h1 = 0.;
  } else if (m1 == 0 && m2 == 1 && m3 == 1 && m4 == 0) {                              // :249-261
    st = ST_0110;
    a32 = a32 - a22;
    h3 = -a22 / a32;
    h2 = 1. - h3;
    h1 = 0.; h4 = 0.;
  } else if (m1 == 1 && m2 == 1) {                                                    // :263-275
    st = ST_1100;
    a22 = a22 - a12;
    h2 = -a12 / a22;
    h1 = 1. - h2;
    h3 = 0.; h4 = 0.;
  } else if (m3 == 1 && m4 == 1) {                                                    // :277-289
    st = ST_0011;
    a42 = a42 - a32;
    h4 = -a32 / a42;
    h3 = 1. - h4;
    h1 = 0.; h2 = 0.;
  } else if (m2 == 1) { st = ST_0100; h1 = 0.; h2 = 1.; h3 = 0.; h4 = 0.; }           // :291-301
  else if (m3 == 1) { st = ST_0010; h1 = 0.; h2 = 0.; h3 = 1.; h4 = 0.; }             // :303-313
  else { st = ST_0000; h1 = 0.; h2 = 0.; h3 = 0.; h4 = 0.; }
  T.stencil[c] = st;
  T.hevc[0][c] = h1; T.hevc[1][c] = h2; T.hevc[2][c] = h3; T.hevc[3][c] = h4;
  // set_slope_coeffs / set_d2_mask on the 3-cell stencil (i-1,i,i+1)
  if (m2 == 0 || m3 == 0 || m4 == 0) { T.ssc[c] = 0.; T.scc[c] = 0.; T.d2m[c] = 0.; }
  else { T.ssc[c] = 2.; T.scc[c] = 2. * dx3 / (dx2 + 2. * dx3 + dx4); T.d2m[c] = 1.; }
}

// stencil tag halo through a real plane, :2612-2614 / :2635-2637
__global__ void k_cppm_tag_convert(const DevView *__restrict__ Vp, int dir, int back) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  (void)i; (void)j;
  gi_t st = dir ? V.m[I_cppm_stj] : V.m[I_cppm_sti];
  if (!back) V.f[F_util1][c] = (double)st[c];
  else st[c] = (int)lround(V.f[F_util1][c]);
}

// With the arctic patch the grid folds onto itself across row jj: what is "left" of a cell there is
// "right" in the mirrored halo, so the one-sided stencil tags and the edge-value coefficients swap
// (phy/mod_cppm.F90:2650-2722).  which = 0: i-direction tables on row jj, i = 1-nbdy..ii+nbdy;
// which = 1: j-direction tables on row jj from the global mid column on, and on rows jj+1..jj+nbdy.
__device__ inline int cppm_mirror_tag(int st) {
  switch (st) {
    case ST_1110: return ST_0111;
    case ST_0111: return ST_1110;
    case ST_1100: return ST_0011;
    case ST_0011: return ST_1100;
    case ST_0100: return ST_0010;
    case ST_0010: return ST_0100;
    default: return st;
  }
}
__global__ void k_cppm_arctic_init_swap(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (V.j0 + V.jj != V.jtdm) return;                        // nproc == jpr
  const int which = by_;
  if (which == 0) {
    if (j != V.jj) return;                                  // i = 1-nbdy..ii+nbdy: the whole padded row
  } else {
    const int ilo = V.itdm / 2 - V.i0 + 1;
    const bool seam = j == V.jj && i >= (ilo > 1 ? ilo : 1) && i <= V.ii;
    const bool beyond = j >= V.jj + 1 && j <= V.jj + NBDY && i >= 1 && i <= V.ii;
    if (!seam && !beyond) return;
  }
  gi_t st = which ? V.m[I_cppm_stj] : V.m[I_cppm_sti];
  st[c] = cppm_mirror_tag(st[c]);
  gd_t h1 = V.f[which ? F_hevc1j : F_hevc1i], h2 = V.f[which ? F_hevc2j : F_hevc2i];
  gd_t h3 = V.f[which ? F_hevc3j : F_hevc3i], h4 = V.f[which ? F_hevc4j : F_hevc4i];
  double t = h1[c]; h1[c] = h4[c]; h4[c] = t;
  t = h2[c]; h2[c] = h3[c]; h3[c] = t;
}

int st_init_cppm(blomgpu_ctx *c) {
  const DevView &h = c->h;
  hipLaunchKernelGGL(k_cppm_init, plane_grid(h, 2), dim3(256), 0, c->stream, c->d);
  for (int dir = 0; dir < 2; dir++) {
    // halo types as in the reference (:2606-2648): u-grid for the i tables, v-grid for the j tables,
    // p-grid for the slope/curvature masks; they only matter across the arctic seam
    const int mh = dir ? 0 : NBDY, nh = dir ? NBDY : 0, it = dir ? 4 : 3;
    hipLaunchKernelGGL(k_cppm_tag_convert, plane_grid(h), dim3(256), 0, c->stream, c->d, dir, 0);
    if (int rc = st_xctilr(c, h.f[F_util1], 1, 1, mh, nh, it)) return rc;
    hipLaunchKernelGGL(k_cppm_tag_convert, plane_grid(h), dim3(256), 0, c->stream, c->d, dir, 1);
    const int edge[] = {dir ? F_hevc1j : F_hevc1i, dir ? F_hevc2j : F_hevc2i, dir ? F_hevc3j : F_hevc3i,
                        dir ? F_hevc4j : F_hevc4i};
    for (int f : edge)
      if (int rc = st_xctilr(c, h.f[f], 1, 1, mh, nh, it)) return rc;
    const int twelve[] = {dir ? F_tmc0j : F_tmc0i, dir ? F_tmclj : F_tmcli, dir ? F_tmcrj : F_tmcri};
    for (int f : twelve)
      if (int rc = st_xctilr(c, h.f[f], 1, 12, mh, nh, it)) return rc;
    const int mask[] = {dir ? F_sscj : F_ssci, dir ? F_sccj : F_scci, dir ? F_d2mj : F_d2mi};
    for (int f : mask)
      if (int rc = st_xctilr(c, h.f[f], 1, 1, mh, nh, 1)) return rc;
  }
  if (h.nreg == 2) hipLaunchKernelGGL(k_cppm_arctic_init_swap, plane_grid(h, 2), dim3(256), 0, c->stream, c->d);
  HIPCHK(c, hipGetLastError());
  c->cppm_ready = true;
  return 0;
}

// thickness edge values next to the arctic seam swap left and right after their halo update
// (i sweeps :1531-1541 / :1848-1858: row jj, i = 1-w..ii+w; j sweeps :1686-1704 / :2002-2020: row jj from
// the global mid column on, rows jj+1..jj+w for i = 1..ii)
template <int DIR>
__global__ void k_cppm_arctic_edge_swap(const DevView *__restrict__ Vp, int w) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (V.j0 + V.jj != V.jtdm) return;
  if (DIR == 0) {
    if (j != V.jj || i < 1 - w || i > V.ii + w) return;
  } else {
    const int ilo = V.itdm / 2 - V.i0 + 1;
    const bool seam = j == V.jj && i >= (ilo > 1 ? ilo : 1) && i <= V.ii;
    const bool beyond = j >= V.jj + 1 && j <= V.jj + w && i >= 1 && i <= V.ii;
    if (!seam && !beyond) return;
  }
  const size_t o = c + (size_t)by_ * V.nplane;
  const double t = V.f[F_hel_3d][o];
  V.f[F_hel_3d][o] = V.f[F_her_3d][o];
  V.f[F_her_3d][o] = t;
}

// ---- sweep kernels, DIR = 0: along i (transport by cau), 1: along j (cav) ---------------------------
// s = coordinate along the sweep, o = the other one; sd = memory stride along the sweep
#define SWEEP_COORDS(V)                                                    \
  const int sd = DIR ? (V).ni : 1, od = DIR ? 1 : (V).ni;                  \
  const int s = DIR ? j : i, o = DIR ? i : j;                              \
  const int sdm = DIR ? (V).jj : (V).ii, odm = DIR ? (V).ii : (V).jj;      \
  (void)od; (void)sd

// LIM = 0: non-oscillatory limiting (reach 4 cells along the sweep), 1: monotonic (reach 3)
template <int DIR, int LIM>
__global__ void k_cppm_hm(const DevView *__restrict__ Vp, int nn, int second_pass) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  SWEEP_COORDS(V);
  if (o < 1 || o > odm || s < -3 + LIM || s > sdm + 4 - LIM) return;
  const int k = by_;
  const size_t np = V.nplane, ok = (size_t)k * np;
  double h = fmax2(0., V.f[F_dp][c + (size_t)(k + nn) * np]) + DPEPS;
  if (second_pass) {                       // divergence of the other direction's Courant number, :1501-1509
    gcd_t ca2 = (DIR ? V.f[F_cau] : V.f[F_cav]) + ok;
    h = h / (1. - (ca2[c + od] - ca2[c]) * V.f[F_scp2i][c]);
  }
  WK(V, W_HM)[c + ok] = h;
}

// Thickness edge values of cell c with limiting: h_edges_nosc :361-434 (LIM 0), h_edges_mono :436-488
// (LIM 1); the same code opens parabola_coeffs_pc_nosc :1144-1207 and parabola_coeffs_pc_mono :1294-1326.
template <int LIM>
__device__ inline void cppm_h_edges(const CppmTab &T, size_t c, int sd, const double *hmv, double &hel, double &her) {
  double hm[7];                                  // hm(s-3..s+3)
#pragma unroll
  for (int q = LIM; q < 7 - LIM; q++) hm[q] = hmv[c + (q - 3) * sd];
#define HM(x) hm[(x) + 3]
  double he[4];                                  // edge values at s-1..s+2 (LIM 1: only s, s+1 are used)
#pragma unroll
  for (int q = LIM; q < 4 - LIM; q++) {
    const int x = q - 1;
    const size_t cx = c + x * sd;
    he[q] = T.hevc[0][cx] * HM(x - 2) + T.hevc[1][cx] * HM(x - 1) + T.hevc[2][cx] * HM(x) + T.hevc[3][cx] * HM(x + 1);
  }
  hel = he[1];
  her = he[2];
  const double hm0 = HM(0), hmm = HM(-1), hmp = HM(1);
  bool limit = true;
  if (LIM == 0) {                                // hel(x) = he(x), her(x) = he(x+1); d2h at s-1, s, s+1
    double d2h[3];
#pragma unroll
    for (int q = 0; q < 3; q++) d2h[q] = T.d2m[c + (q - 1) * sd] * (he[q] - 2. * HM(q - 1) + he[q + 1]);
    limit = d2h[0] * d2h[1] <= 0. || d2h[1] * d2h[2] <= 0.;
  }
  if (limit) {
    const double ssc = T.ssc[c];
    const double sl = ssc * (hm0 - hmm), sr = ssc * (hmp - hm0);
    if (sl * sr > 0.) {
      double sc = T.scc[c] * (hmp - hmm);
      sc = copysign(fmin2(fmin2(fabs(sl), fabs(sr)), fabs(sc)), sc);
      if ((hmm - hel) * (hm0 - hel) > 0.) hel = hm0 - copysign(fmin2(.5 * fabs(sc), fabs(hel - hm0)), sc);
      if ((hmp - her) * (hm0 - her) > 0.) her = hm0 + copysign(fmin2(.5 * fabs(sc), fabs(her - hm0)), sc);
      const double d = her - hel;
      const double q = d * (2. * hm0 - hel - her);
      const double r = (1. / 3.) * d * d;
      if (q > r) hel = 3. * hm0 - 2. * her;
      else if (-r > q) her = 3. * hm0 - 2. * hel;
    } else {
      hel = hm0;
      her = hm0;
    }
  }
  if (LIM == 0) {
    hel = fmax2(hel, DPEPS);
    her = fmax2(her, DPEPS);
    const double sl = 2. * (3. * hm0 - 2. * hel - her);
    const double a2 = 3. * (hel - 2. * hm0 + her);
    const double sr = sl + 2. * a2;
    if (sl < 0. && sr > 0.) {
      if (a2 * hel - .25 * sl * sl < a2 * DPEPS) {
        const double q = 3. * hm0 / (3. * sl * sr + 4. * a2 * a2);
        hel = sl * sl * q;
        her = sr * sr * q;
      }
    }
  }
#undef HM
}

template <int DIR, int LIM>
__global__ void k_cppm_hedges(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  SWEEP_COORDS(V);
  if (o < 1 || o > odm || s < 1 || s > sdm) return;
  const size_t ok = (size_t)by_ * V.nplane;
  const CppmTab T = cppm_tab(V, DIR);
  double hel, her;
  cppm_h_edges<LIM>(T, c, sd, WK(V, W_HM) + ok, hel, her);
  V.f[F_hel_3d][c + ok] = hel;
  V.f[F_her_3d][c + ok] = her;
}

// tracer edge value coefficients of one edge, parabola_coeffs_fc_nosc :520-731 == parabola_coeffs_fc_mono :845-1050
__device__ inline void cppm_tevc(const CppmTab &T, size_t np, size_t cx, int sd, const double *hm, const double *hel,
                                 const double *her, double &t1, double &t2, double &t3, double &t4) {
#define TM0(q) T.tmc0[cx + (size_t)((q)-1) * np]
#define TML(q) T.tmcl[cx + (size_t)((q)-1) * np]
#define TMR(q) T.tmcr[cx + (size_t)((q)-1) * np]
#define AT_(a, d) (a)[cx + (d) * sd]
  double h1i, h2i, h3i, h4i, a12, a13, a14, a22, a23, a24, a32, a33, a34, a42, a43, a44, q;
  switch (T.stencil[cx]) {
    case ST_1111:
      h1i = 1. / AT_(hm, -2); h2i = 1. / AT_(hm, -1); h3i = 1. / AT_(hm, 0); h4i = 1. / AT_(hm, 1);
      a12 = TM0(1) + (TML(1) * AT_(hel, -2) + TMR(1) * AT_(her, -2)) * h1i;
      a13 = TM0(2) + (TML(2) * AT_(hel, -2) + TMR(2) * AT_(her, -2)) * h1i;
      a14 = TM0(3) + (TML(3) * AT_(hel, -2) + TMR(3) * AT_(her, -2)) * h1i;
      a22 = TM0(4) + (TML(4) * AT_(hel, -1) + TMR(4) * AT_(her, -1)) * h2i - a12;
      a23 = TM0(5) + (TML(5) * AT_(hel, -1) + TMR(5) * AT_(her, -1)) * h2i - a13;
      a24 = TM0(6) + (TML(6) * AT_(hel, -1) + TMR(6) * AT_(her, -1)) * h2i - a14;
      a32 = TM0(7) + (TML(7) * AT_(hel, 0) + TMR(7) * AT_(her, 0)) * h3i - a12;
      a33 = TM0(8) + (TML(8) * AT_(hel, 0) + TMR(8) * AT_(her, 0)) * h3i - a13;
      a34 = TM0(9) + (TML(9) * AT_(hel, 0) + TMR(9) * AT_(her, 0)) * h3i - a14;
      a42 = TM0(10) + (TML(10) * AT_(hel, 1) + TMR(10) * AT_(her, 1)) * h4i - a12;
      a43 = TM0(11) + (TML(11) * AT_(hel, 1) + TMR(11) * AT_(her, 1)) * h4i - a13;
      a44 = TM0(12) + (TML(12) * AT_(hel, 1) + TMR(12) * AT_(her, 1)) * h4i - a14;
      q = 1. / a22;
      a23 = a23 * q;
      a33 = a33 - a23 * a32;
      a43 = a43 - a23 * a42;
      a24 = a24 * q;
      a34 = a34 - a24 * a32;
      a44 = a44 - a24 * a42;
      a34 = a34 / a33;
      a44 = a44 - a34 * a43;
      t2 = -a12;
      t3 = -a13 - a23 * t2;
      t4 = -a14 - a24 * t2 - a34 * t3;
      t4 = t4 / a44;
      t3 = (t3 - a43 * t4) / a33;
      t2 = (t2 - a32 * t3 - a42 * t4) / a22;
      t1 = 1. - t2 - t3 - t4;
      break;
    case ST_1110:
      h1i = 1. / AT_(hm, -2); h2i = 1. / AT_(hm, -1); h3i = 1. / AT_(hm, 0);
      a12 = TM0(1) + (TML(1) * AT_(hel, -2) + TMR(1) * AT_(her, -2)) * h1i;
      a13 = TM0(2) + (TML(2) * AT_(hel, -2) + TMR(2) * AT_(her, -2)) * h1i;
      a22 = TM0(4) + (TML(4) * AT_(hel, -1) + TMR(4) * AT_(her, -1)) * h2i - a12;
      a23 = TM0(5) + (TML(5) * AT_(hel, -1) + TMR(5) * AT_(her, -1)) * h2i - a13;
      a32 = TM0(7) + (TML(7) * AT_(hel, 0) + TMR(7) * AT_(her, 0)) * h3i - a12;
      a33 = TM0(8) + (TML(8) * AT_(hel, 0) + TMR(8) * AT_(her, 0)) * h3i - a13;
      a23 = a23 / a22;
      a33 = a33 - a23 * a32;
      t2 = -a12;
      t3 = -a13 - a23 * t2;
      t3 = t3 / a33;
      t2 = (t2 - a32 * t3) / a22;
      t1 = 1. - t2 - t3;
      t4 = 0.;
      break;
    case ST_0111:
      h2i = 1. / AT_(hm, -1); h3i = 1. / AT_(hm, 0); h4i = 1. / AT_(hm, 1);
      a22 = TM0(4) + (TML(4) * AT_(hel, -1) + TMR(4) * AT_(her, -1)) * h2i;
      a23 = TM0(5) + (TML(5) * AT_(hel, -1) + TMR(5) * AT_(her, -1)) * h2i;
      a32 = TM0(7) + (TML(7) * AT_(hel, 0) + TMR(7) * AT_(her, 0)) * h3i - a22;
      a33 = TM0(8) + (TML(8) * AT_(hel, 0) + TMR(8) * AT_(her, 0)) * h3i - a23;
      a42 = TM0(10) + (TML(10) * AT_(hel, 1) + TMR(10) * AT_(her, 1)) * h4i - a22;
      a43 = TM0(11) + (TML(11) * AT_(hel, 1) + TMR(11) * AT_(her, 1)) * h4i - a23;
      a33 = a33 / a32;
      a43 = a43 - a33 * a42;
      t3 = -a22;
      t4 = -a23 - a33 * t3;
      t4 = t4 / a43;
      t3 = (t3 - a42 * t4) / a32;
      t2 = 1. - t3 - t4;
      t1 = 0.;
      break;
    case ST_1100:
      h1i = 1. / AT_(hm, -2); h2i = 1. / AT_(hm, -1);
      a12 = TM0(1) + (TML(1) * AT_(hel, -2) + TMR(1) * AT_(her, -2)) * h1i;
      a22 = TM0(4) + (TML(4) * AT_(hel, -1) + TMR(4) * AT_(her, -1)) * h2i - a12;
      t2 = -a12 / a22;
      t1 = 1. - t2;
      t3 = 0.; t4 = 0.;
      break;
    case ST_0110:
      h2i = 1. / AT_(hm, -1); h3i = 1. / AT_(hm, 0);
      a22 = TM0(4) + (TML(4) * AT_(hel, -1) + TMR(4) * AT_(her, -1)) * h2i;
      a32 = TM0(7) + (TML(7) * AT_(hel, 0) + TMR(7) * AT_(her, 0)) * h3i - a22;
      t3 = -a22 / a32;
      t2 = 1. - t3;
      t1 = 0.; t4 = 0.;
      break;
    case ST_0011:
      h3i = 1. / AT_(hm, 0); h4i = 1. / AT_(hm, 1);
      a32 = TM0(7) + (TML(7) * AT_(hel, 0) + TMR(7) * AT_(her, 0)) * h3i;
      a42 = TM0(10) + (TML(10) * AT_(hel, 1) + TMR(10) * AT_(her, 1)) * h4i - a32;
      t4 = -a32 / a42;
      t3 = 1. - t4;
      t1 = 0.; t2 = 0.;
      break;
    case ST_0100: t1 = 0.; t2 = 1.; t3 = 0.; t4 = 0.; break;
    case ST_0010: t1 = 0.; t2 = 0.; t3 = 1.; t4 = 0.; break;
    default: t1 = 0.; t2 = 0.; t3 = 0.; t4 = 0.; break;
  }
#undef TM0
#undef TML
#undef TMR
#undef AT_
}

// tracer nt of the sweep: 0 temp, 1 saln, 2.. trc(nt-2)  (tm(1), tm(2), tm(3..ntr_loc) of the reference)
__device__ inline double *cppm_tracer(const DevView &V, int nt, int k, int nn) {
  const size_t np = V.nplane;
  if (nt == 0) return V.f[F_temp] + (size_t)(k + nn) * np;
  if (nt == 1) return V.f[F_saln] + (size_t)(k + nn) * np;
  return V.f[F_trc] + ((size_t)(k + nn) + (size_t)(nt - 2) * 2 * V.kk) * np;
}

// tracer edge values at every edge.  FC = 1: coefficients compatible with the thickness reconstruction
// (4x4 LU, above); FC = 0 ('partial'): the thickness edge coefficients hevc (:1146-1157, :1296-1307)
template <int DIR, int LIM, int FC>
__global__ void k_cppm_tedge(const DevView *__restrict__ Vp, int nn, int ntl) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  SWEEP_COORDS(V);
  if (o < 1 || o > odm || s < -1 + LIM || s > sdm + 3 - LIM) return;
  const int k = by_;
  const size_t np = V.nplane, ok = (size_t)k * np;
  const CppmTab T = cppm_tab(V, DIR);
  double t1, t2, t3, t4;
  if (FC) cppm_tevc(T, np, c, sd, WK(V, W_HM) + ok, V.f[F_hel_3d] + ok, V.f[F_her_3d] + ok, t1, t2, t3, t4);
  else { t1 = T.hevc[0][c]; t2 = T.hevc[1][c]; t3 = T.hevc[2][c]; t4 = T.hevc[3][c]; }
  for (int nt = 0; nt < ntl; nt++) {
    const double *tm = cppm_tracer(V, nt, k, nn);
    WK(V, W_TE + nt)[c + ok] = t1 * tm[c - 2 * sd] + t2 * tm[c - sd] + t3 * tm[c] + t4 * tm[c + sd];
  }
}

// limiting and parabola coefficients: parabola_coeffs_fc_nosc :733-816, _fc_mono :1062-1114,
// _pc_nosc :1159-1262, _pc_mono :1309-1369
template <int DIR, int LIM, int FC>
__global__ void k_cppm_parab(const DevView *__restrict__ Vp, int nn, int ntl) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  SWEEP_COORDS(V);
  if (o < 1 || o > odm || s < 0 || s > sdm + 1) return;
  const int k = by_;
  const size_t np = V.nplane, ok = (size_t)k * np;
  const CppmTab T = cppm_tab(V, DIR);
  gcd_t hmv = WK(V, W_HM) + ok;
  const double hm0 = hmv[c];
  double hel0, her0;
  // thickness factors of the tracer parabolas (FC) at s-1, s, s+1; for 'partial' they are constants
  double hf2m[3], hf2l[3], hf2r[3], d2mv[3];
  double hf1m = 6., hf1l = -4., hf1r = -2.;
  if (FC) {
    gcd_t helv = V.f[F_hel_3d] + ok, herv = V.f[F_her_3d] + ok;
    hel0 = helv[c];
    her0 = herv[c];
#pragma unroll
    for (int q = LIM; q < 3 - LIM; q++) {                                             // :735-750, :1064-1071
      const size_t cx = c + (q - 1) * sd;
      const double hm = hmv[cx], hel = helv[cx], her = herv[cx];
      const double qq = 1. / (12. * hm - hel - her);
      const double f1m = 60. * hm * qq;
      hf2m[q] = -f1m;
      hf2l[q] = 5. * (6. * hm + hel - her) * qq;
      hf2r[q] = 5. * (6. * hm - hel + her) * qq;
      if (q == 1) {
        hf1m = f1m;
        hf1l = -(42. * hm + 4. * hel - 6. * her) * qq;
        hf1r = -(18. * hm - 4. * hel + 6. * her) * qq;
      }
    }
  } else {
    cppm_h_edges<LIM>(T, c, sd, hmv, hel0, her0);
#pragma unroll
    for (int q = 0; q < 3; q++) { hf2m[q] = -2.; hf2l[q] = 1.; hf2r[q] = 1.; }        // d2t = d2m*(tel - 2 tm + ter)
  }
  if (LIM == 0) {
#pragma unroll
    for (int q = 0; q < 3; q++) d2mv[q] = T.d2m[c + (q - 1) * sd];
  }
  const double ssc = T.ssc[c], scc = T.scc[c];
  for (int nt = 0; nt < ntl; nt++) {
    const double *tm = cppm_tracer(V, nt, k, nn), *te = WK(V, W_TE + nt) + ok;
    const double tmm = tm[c - sd], tm0 = tm[c], tmp = tm[c + sd];
    double tel = te[c], ter = te[c + sd];
    bool limit = true;
    if (LIM == 0) {
      double d2t[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const size_t cx = c + (q - 1) * sd;
        if (FC) d2t[q] = d2mv[q] * (hf2m[q] * tm[cx] + hf2l[q] * te[cx] + hf2r[q] * te[cx + sd]);
        else d2t[q] = d2mv[q] * (te[cx] - 2. * tm[cx] + te[cx + sd]);
      }
      limit = d2t[0] * d2t[1] <= 0. || d2t[1] * d2t[2] <= 0.;
    }
    if (limit) {
      double sl = ssc * (tm0 - tmm), sr = ssc * (tmp - tm0);
      if (sl * sr > 0.) {
        double sc = scc * (tmp - tmm);
        sc = copysign(fmin2(fmin2(fabs(sl), fabs(sr)), fabs(sc)), sc);
        if ((tmm - tel) * (tm0 - tel) > 0.) tel = tm0 - copysign(fmin2(.5 * fabs(sc), fabs(tel - tm0)), sc);
        if ((tmp - ter) * (tm0 - ter) > 0.) ter = tm0 + copysign(fmin2(.5 * fabs(sc), fabs(ter - tm0)), sc);
        if (FC) {
          sl = hf1m * tm0 + hf1l * tel + hf1r * ter;
          const double a2 = hf2m[1] * tm0 + hf2l[1] * tel + hf2r[1] * ter;
          sr = sl + 2. * a2;
          if (sl * sr < 0.) {
            if ((ter - tel) * a2 < 0.)
              tel = -((hf1m + 2. * hf2m[1]) * tm0 + (hf1r + 2. * hf2r[1]) * ter) / (hf1l + 2. * hf2l[1]);
            else
              ter = -(hf1m * tm0 + hf1l * tel) / hf1r;
          }
        } else {
          const double d = ter - tel;
          const double q = d * (2. * tm0 - tel - ter);
          const double r = (1. / 3.) * d * d;
          if (q > r) tel = 3. * tm0 - 2. * ter;
          else if (-r > q) ter = 3. * tm0 - 2. * tel;
        }
      } else {
        tel = tm0;
        ter = tm0;
      }
    }
    if (LIM == 0 && nt >= 1) {                                                        // nt = 2..ntr_loc: positive definite
      tel = fmax2(tel, 0.);
      ter = fmax2(ter, 0.);
      double sl, a2;
      if (FC) {
        sl = hf1m * tm0 + hf1l * tel + hf1r * ter;
        a2 = hf2m[1] * tm0 + hf2l[1] * tel + hf2r[1] * ter;
      } else {
        sl = 2. * (3. * tm0 - 2. * tel - ter);
        a2 = 3. * (tel - 2. * tm0 + ter);
      }
      const double sr = sl + 2. * a2;
      if (sl < 0. && sr > 0.) {
        if (a2 * tel - .25 * sl * sl < 0.) {
          const double q = 3. * tm0 / (3. * sl * sr + 4. * a2 * a2);
          tel = sl * sl * q;
          ter = sr * sr * q;
        }
      }
    }
    WK(V, W_TPC0(ntl) + nt)[c + ok] = tel;
    if (FC) {
      WK(V, W_TPC1(ntl) + nt)[c + ok] = hf1m * tm0 + hf1l * tel + hf1r * ter;
      WK(V, W_TPC2(ntl) + nt)[c + ok] = hf2m[1] * tm0 + hf2l[1] * tel + hf2r[1] * ter;
    } else {
      WK(V, W_TPC1(ntl) + nt)[c + ok] = 6. * tm0 - 4. * tel - 2. * ter;
      WK(V, W_TPC2(ntl) + nt)[c + ok] = 3. * (tel - 2. * tm0 + ter);
    }
  }
  WK(V, W_HPC0)[c + ok] = hel0;
  WK(V, W_HPC1)[c + ok] = 6. * hm0 - 4. * hel0 - 2. * her0;
  WK(V, W_HPC2)[c + ok] = 3. * (hel0 - 2. * hm0 + her0);
}

// flux_integration :1373-1468 + flux accumulation :1612-1618
template <int DIR>
__global__ void k_cppm_flux(const DevView *__restrict__ Vp, int n, int mm, int ntl) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  SWEEP_COORDS(V);
  if (o < 1 || o > odm || s < 1 || s > sdm + 1) return;
  const int k = by_;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np;
  const double c1_2 = 1. / 2., c1_3 = 1. / 3., c1_4 = 1. / 4., c1_5 = 1. / 5.;
  const double ca = (DIR ? V.f[F_cav] : V.f[F_cau])[c + ok];
  const double db = (DIR ? V.f[F_pbv] : V.f[F_pbu])[c + (size_t)(n - 1) * np];
  gcd_t ai = V.f[F_scp2i], p = V.f[F_p];
  gcd_t hpc0 = WK(V, W_HPC0) + ok, hpc1 = WK(V, W_HPC1) + ok, hpc2 = WK(V, W_HPC2) + ok;
  double hf, p0, p1, p2;
  size_t up;
  if (ca < 0.) {
    up = c;
    const double cc = ca * ai[c];
    if (p[c + (size_t)(k + 1) * np] > db) {
      const double hb = fmax2(0., db - p[c + ok]);
      hf = hb * ca;
      p0 = hb;
      p1 = -c1_2 * hb * cc;
      p2 = c1_3 * hb * cc * cc;
    } else {
      const double h0 = hpc0[c], h1 = hpc1[c], h2 = hpc2[c];
      hf = (h0 - (c1_2 * h1 - c1_3 * h2 * cc) * cc) * ca;
      p0 = h0 - (c1_2 * h1 - c1_3 * h2 * cc) * cc;
      p1 = -(c1_2 * h0 - (c1_3 * h1 - c1_4 * h2 * cc) * cc) * cc;
      p2 = (c1_3 * h0 - (c1_4 * h1 - c1_5 * h2 * cc) * cc) * cc * cc;
    }
  } else {
    up = c - sd;
    const double cc = ca * ai[up];
    const double q1 = 1. - c1_2 * cc;
    const double q2 = 1. - (1. - c1_3 * cc) * cc;
    if (p[up + (size_t)(k + 1) * np] > db) {
      const double hb = fmax2(0., db - p[up + ok]);
      hf = hb * ca;
      p0 = hb;
      p1 = q1 * hb;
      p2 = q2 * hb;
    } else {
      const double h0 = hpc0[up], h1 = hpc1[up], h2 = hpc2[up];
      hf = (h0 + q1 * h1 + q2 * h2) * ca;
      const double q3 = c1_4 * (1. + 3. * (1. - cc) * q2);
      const double q4 = c1_5 * (1. + 4. * (1. - cc) * q3);
      p0 = h0 + q1 * h1 + q2 * h2;
      p1 = q1 * h0 + q2 * h1 + q3 * h2;
      p2 = q2 * h0 + q3 * h1 + q4 * h2;
    }
  }
  WK(V, W_HF)[c + ok] = hf;
  gd_t mflx = DIR ? V.f[F_vflx] : V.f[F_uflx], tflx = DIR ? V.f[F_vtflx] : V.f[F_utflx];
  gd_t sflx = DIR ? V.f[F_vsflx] : V.f[F_usflx];
  mflx[c + okm] = mflx[c + okm] + hf;
  for (int nt = 0; nt < ntl; nt++) {
    const double htf = (p0 * WK(V, W_TPC0(ntl) + nt)[up + ok] + p1 * WK(V, W_TPC1(ntl) + nt)[up + ok] +
                        p2 * WK(V, W_TPC2(ntl) + nt)[up + ok]) * ca;
    WK(V, W_HTF(ntl) + nt)[c + ok] = htf;
    if (nt == 0) tflx[c + okm] = tflx[c + okm] + htf;
    if (nt == 1) sflx[c + okm] = sflx[c + okm] + htf;
  }
}

// update with flux divergences, :1597-1610
template <int DIR>
__global__ void k_cppm_update(const DevView *__restrict__ Vp, int nn, int ntl) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  SWEEP_COORDS(V);
  if (o < 1 || o > odm || s < 1 || s > sdm) return;
  const int k = by_;
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np;
  const double ai = V.f[F_scp2i][c];
  const double ho = fmax2(0., V.f[F_dp][c + okn]) + DPEPS;
  gcd_t hf = WK(V, W_HF) + ok;
  const double hn = ho - (hf[c + sd] - hf[c]) * ai;
  const double hni = 1. / hn;
  for (int nt = 0; nt < ntl; nt++) {
    double *tm = cppm_tracer(V, nt, k, nn);
    gcd_t htf = WK(V, W_HTF(ntl) + nt) + ok;
    tm[c] = (ho * tm[c] - (htf[c + sd] - htf[c]) * ai) * hni;
  }
  V.f[F_dp][c + okn] = fmax2(0., hn - DPEPS);
}

template <int DIR, int LIM, int FC>
static int cppm_sweep(blomgpu_ctx *c, int n, int mm, int nn, int k1n, bool second_pass) {
  const DevView &h = c->h;
  const size_t np = h.nplane;
  const int ntl = 2 + h.ntr, w = 4 - LIM, mh = DIR ? 0 : w, nh = DIR ? w : 0;
  const dim3 g = plane_grid(h, h.kk), b(256);
  if (int rc = st_xctilr(c, h.f[F_dp] + (size_t)(k1n - 1) * np, 1, h.kk, mh, nh, 1)) return rc;           // :1484-1489
  if (int rc = st_xctilr(c, h.f[F_temp] + (size_t)(k1n - 1) * np, 1, h.kk, mh, nh, 1)) return rc;
  if (int rc = st_xctilr(c, h.f[F_saln] + (size_t)(k1n - 1) * np, 1, h.kk, mh, nh, 1)) return rc;
  for (int nt = 0; nt < h.ntr; nt++)
    if (int rc = st_xctilr(c, h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np, 1, h.kk, mh, nh, 1)) return rc;
  hipLaunchKernelGGL((k_cppm_hm<DIR, LIM>), g, b, 0, c->stream, c->d, nn, second_pass ? 1 : 0);
  if (FC) {
    hipLaunchKernelGGL((k_cppm_hedges<DIR, LIM>), g, b, 0, c->stream, c->d);
    if (int rc = st_xctilr(c, h.f[F_hel_3d], 1, h.kk, mh, nh, 1)) return rc;                                  // :1527-1528
    if (int rc = st_xctilr(c, h.f[F_her_3d], 1, h.kk, mh, nh, 1)) return rc;
    if (h.nreg == 2) hipLaunchKernelGGL((k_cppm_arctic_edge_swap<DIR>), g, b, 0, c->stream, c->d, w);
  }
  hipLaunchKernelGGL((k_cppm_tedge<DIR, LIM, FC>), g, b, 0, c->stream, c->d, nn, ntl);
  hipLaunchKernelGGL((k_cppm_parab<DIR, LIM, FC>), g, b, 0, c->stream, c->d, nn, ntl);
  hipLaunchKernelGGL((k_cppm_flux<DIR>), g, b, 0, c->stream, c->d, n, mm, ntl);
  hipLaunchKernelGGL((k_cppm_update<DIR>), g, b, 0, c->stream, c->d, nn, ntl);
  HIPCHK(c, hipGetLastError());
  return 0;
}

template <int LIM, int FC>
static int cppm_variant(blomgpu_ctx *c, int n, int mm, int nn, int k1n) {
  const DevView &h = c->h;
  const int w = 4 - LIM;
  if (int rc = st_xctilr(c, h.f[F_cau], 1, h.kk, w, w, 13)) return rc;                  // :2761-2762, :2779-2780
  if (int rc = st_xctilr(c, h.f[F_cav], 1, h.kk, w, w, 14)) return rc;
  if (h.P.nstep % 2 == 1) {                                                             // :2764-2774
    if (int rc = cppm_sweep<0, LIM, FC>(c, n, mm, nn, k1n, false)) return rc;
    return cppm_sweep<1, LIM, FC>(c, n, mm, nn, k1n, true);
  }
  if (int rc = cppm_sweep<1, LIM, FC>(c, n, mm, nn, k1n, false)) return rc;
  return cppm_sweep<0, LIM, FC>(c, n, mm, nn, k1n, true);
}

// cppm, :2748-2834
int st_cppm(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)k1m;
  const DevView &h = c->h;
  if (!c->cppm_ready) return ctx_fail(c, "cppm: init_cppm has not been called (blomgpu_init_cppm)");
  if (W_NSLOT(2 + h.ntr) > h.nwk) return ctx_fail(c, "cppm: device work space too small for this many tracers");
  const bool fc = c->cppm_compat == 1, mono = c->cppm_limiting == 1;
  if (fc) return mono ? cppm_variant<1, 1>(c, n, mm, nn, k1n) : cppm_variant<0, 1>(c, n, mm, nn, k1n);
  return mono ? cppm_variant<1, 0>(c, n, mm, nn, k1n) : cppm_variant<0, 0>(c, n, mm, nn, k1n);
}
