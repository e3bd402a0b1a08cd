// remap: gradient and flux sweep of a layer in one LDS-tiled kernel -- phy/mod_remap.F90:358-1462.
//
// The one-kernel-per-sweep form (stage_advect.hip) writes 8 + 3 ntr gradient planes per layer to HBM and its flux sweep
// fetches them 2.9 times over (donor cells are the point's and its neighbours': 3.0 GB per launch); both sweeps are
// chains of dependent loads (masks -> neighbour indices -> values; corner velocities -> donor cell -> gradients), ~140
// exposed memory latencies per wavefront.  Here a workgroup owns a tile of RT_TW x RT_TH points of one layer:
//   phase 0  every global load of the workgroup, all independent and issued up front: the scalars (dp, p, T, S, tracers)
//            and masks of the tile with a 2-point rim go to LDS; each thread keeps what its own point needs later
//            (flux areas, metrics, face pressures, old fluxes) in registers; non-dimensional face velocities cu, cv of
//            the tile with a 1-point rim go to LDS                                              (mod_remap:588-610)
//   phase 1  one thread per point of the tile with a 1-point rim: limited gradients from the LDS scalars and the
//            corner velocities from cu, cv, into registers                                      (:358-584, :623-659)
//   phase 2  these replace the scalars in LDS
//   phase 3  one thread per point of the tile: u- and v-face flux polygons, donor cells read from LDS      (:661-1462)
// The gradient planes never reach HBM; the price is the rim, 1.33 gradient evaluations per point.
// Expressions are those of k_remap_grad / k_remap_flux (stage_advect.hip), operator for operator: every neighbour value
// selected there by a wet-restricted index is selected here by the same index into the tile.
// Roofline: HBM, ~(19 + 4 ntr) F per launch (scalars with rim 1.7 x (4 + ntr) F, cau, cav, 6 old/new fluxes, 2 (3+ntr) F
// of flux planes for k_remap_update).
#include "remap_common.h"

#ifndef RT_TH                             // (tile height / thread count overridable for measurement builds: 16 / 1024 was tried)
#define RT_TH 8
#define RT_NT 512                         // threads of a workgroup: >= RT_SN, and two per point of the tile
#endif
#define RT_TW 32
#define RT_GW (RT_TW + 2)                 // gradient region: the tile and a 1-point rim
#define RT_GN (RT_GW * (RT_TH + 2))
#define RT_SW (RT_TW + 4)                 // scalar region: the tile and a 2-point rim
#define RT_SN (RT_SW * (RT_TH + 4))
#define RT_NB 4                           // tracers of a later batch (MORE)
#define RT_NSC(ntr) (4 + (ntr))           // scalars: dp, p(k+1), T, S, tracers
#define RT_NG(ntr) (10 + 3 * (ntr))       // gradient slots (remap_common.h) + dp', pup of the cell
#define RT_G_DPT(ntr) (8 + 3 * (ntr))
#define RT_G_PUP(ntr) (9 + 3 * (ntr))

#define MP(m) ((m) & 1)
#define MU(m) (((m) >> 1) & 1)
#define MV(m) (((m) >> 2) & 1)

static_assert(RT_NT >= RT_SN && RT_NT == 2 * RT_TW * RT_TH, "thread count of k_remap_tile");

// The largest / smallest of the eight neighbours with the hardware's v_max_f64 / v_min_f64: one instruction a pair where the
// select form of the reference's max / min (fmax2: compare and two 32-bit selects) takes three, 28 fewer per gradient in a kernel
// bound by instruction issue.  The two forms differ only in which zero they return for a tie of +0 and -0 (no NaN reaches here),
// and the result is used in one place, fmax2(0, max8 - fc) resp. fmin2(0, min8 - fc): a zero of either sign gives a zero of
// some sign there, the limiter's test (> 0, < 0) fails for both and the gradient is set to zero -- no bit of the output depends
// on it.  (Inline assembly: the compiler's own maxnum brings a canonicalising instruction per operand with it.)
// The same instructions for the limiter's other maxima and minima (hmax2 / hmin2), where the same holds -- checked case by case:
//   max(q1, q2) + max(q3, q4) and the minima: the sums are only compared with tfmx > 0 resp. tfmn < 0 (a zero of either sign loses);
//   max(tfmx, tgmx), min(tfmn, tgmn) with tfmx > 0 > tfmn: never a tie of zeros; min of the two quotients: both in (0, 1];
//   max(0, dp) + dpeps: either zero plus dpeps is dpeps; max(dpeps, min(pbmin - pup, dp')): a zero of either sign gives dpeps.
#ifdef BLOM_HOSTEMU
#define hmax8 max8
#define hmin8 min8
#define hmax2 fmax2
#define hmin2 fmin2
#else
#define hmax2 hw_max
#define hmin2 hw_min
__device__ inline double hw_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline double hw_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline double hmax8(double a, double b, double c, double d, double e, double f, double g, double h) {
  return hw_max(hw_max(hw_max(a, b), hw_max(c, d)), hw_max(hw_max(e, f), hw_max(g, h)));
}
__device__ inline double hmin8(double a, double b, double c, double d, double e, double f, double g, double h) {
  return hw_min(hw_min(hw_min(a, b), hw_min(c, d)), hw_min(hw_min(e, f), hw_min(g, h)));
}
#endif

// limited gradient of one scalar from its tile, mod_remap.F90:412-439 (limited_gradient of stage_advect.hip)
struct TNbr {
  int w, e, s, n, sw, se, nw, ne;
};
__device__ inline void limited_gradient_t(const double *f, const TNbr &b, int c, double dxi, double dyi, double xd, double yd,
                                          double &gx, double &gy, double &gd) {
  const double fc = f[c], fw = f[b.w], fe = f[b.e], fs = f[b.s], fn = f[b.n];
  double tx = (fe - fw) * dxi;
  double ty = (fn - fs) * dyi;
  const double q1 = tx * (-.5 - xd), q2 = tx * (.5 - xd), q3 = ty * (-.5 - yd), q4 = ty * (.5 - yd);
  const double tgmx = hmax2(q1, q2) + hmax2(q3, q4);
  const double tgmn = hmin2(q1, q2) + hmin2(q3, q4);
  const double fsw = f[b.sw], fse = f[b.se], fnw = f[b.nw], fne = f[b.ne];
  const double tfmx = hmax2(0., hmax8(fsw, fs, fse, fw, fe, fnw, fn, fne) - fc);
  const double tfmn = hmin2(0., hmin8(fsw, fs, fse, fw, fe, fnw, fn, fne) - fc);
  if (tfmx > 0. && tfmn < 0.) {
    const double q = hmin2(tfmx / hmax2(tfmx, tgmx), tfmn / hmin2(tfmn, tgmn));
    tx = tx * q;
    ty = ty * q;
    gd = fc - tx * xd - ty * yd;
  } else {
    tx = 0.;
    ty = 0.;
    gd = fc;
  }
  gx = tx;
  gy = ty;
}

// what a face keeps of one polygon for the tracers of later batches: the mass-flux contribution and the two moments, and
// the donor cell
struct PolyRec {
  double fd, qx, qy;
  int x;
};
struct AdvList {               // model indices of the advected tracers, in order (ints: an entry at a position the workgroup computes is
  int idx[64];                 // a scalar load from the kernel arguments; bytes were vector loads, a memory round trip in front of every
};                             // batch's loads and stores)
// one polygon's contribution from the donor cell at index x of the gradient region (add_contrib of stage_advect.hip)
// (rec[slot]: slot 0 / 1 the triangles at the face's two ends where there is one, slot 2 the pentagon -- the order they are added in)
template <bool MORE>
__device__ inline void add_contrib_t(const double *g, int ntr, int x, double pbface, double a, double ax, double ay, double axx,
                                     double ayy, double axy, Acc &A, PolyRec *rec, const int slot) {
  const double dpt = g[RT_G_DPT(ntr) * RT_GN + x];
  const double pup = g[RT_G_PUP(ntr) * RT_GN + x];
  const double dl = fmin2(dpt, fmax2(0., pbface - pup));
  const double dx = g[G_DX * RT_GN + x], dy = g[G_DY * RT_GN + x];
  const double fd = a * dl + ax * dx + ay * dy;
  A.fd = A.fd + fd;
  const double qx = ax * dl + axx * dx + axy * dy;
  const double qy = ay * dl + axy * dx + ayy * dy;
  A.ft = A.ft + fd * g[G_TD * RT_GN + x] + qx * g[G_TX * RT_GN + x] + qy * g[G_TY * RT_GN + x];
  A.fs = A.fs + fd * g[G_SD * RT_GN + x] + qx * g[G_SX * RT_GN + x] + qy * g[G_SY * RT_GN + x];
#pragma unroll
  for (int nt = 0; nt < MAXTR; nt++)
    if (nt < ntr)
      A.ftr[nt] = A.ftr[nt] + fd * g[G_TRD(nt) * RT_GN + x] + qx * g[G_TRX(nt) * RT_GN + x] + qy * g[G_TRY(nt) * RT_GN + x];
  if (MORE) { rec[slot].fd = fd; rec[slot].qx = qx; rec[slot].qy = qy; rec[slot].x = x; }
}

// MORE = true (more than MAXTR advected tracers): the first MAXTR ride with dp, T, S as always; the others follow in batches
// of MAXTR through the same LDS slots -- their scalars in, their limited gradients (the wet-restricted neighbours, dxi, dyi
// and the centre-of-mass offsets of the point kept in registers), their fluxes from the polygons' mass-flux contribution
// and moments each face recorded (PolyRec): the geometry is evaluated once for all tracers.
// FOLD = true (inside blomgpu_step): the flux-divergence update of k_remap_update (mod_remap.F90:1468-1520) is done here.  A
// cell needs the fluxes through its east and north face as well, which are the west and south face of its neighbours: the
// tiles are laid out RT_TW-1 x RT_TH-1 apart, a workgroup evaluates the faces of all its RT_TW x RT_TH points but OWNS only
// those with a neighbour to the east and to the north inside the tile -- for these it accumulates uflx.. and writes the new
// dp, T, S and tracers.  Not in place (the neighbouring tiles still read the old values in their rims): into work-space
// planes, where pbcor1 -- the next stage, and the only reader before diffus rewrites the fields -- takes them
// (remap_common.h: R_DP..).  The 12 + 2 ntr flux planes between the two kernels never reach memory.
template <bool MORE, bool FOLD>
__global__ void __launch_bounds__(RT_NT, 4) k_remap_tile(const DevView *__restrict__ Vp, int n, int mm, int nn, int ntx, int nadv_all, int nfirst, AdvList L, int tsel KPROF_ARGS) {
  const DevView &V = *Vp;
  const int nadv = nfirst;               // tracers that ride with dp, T, S (all of them unless MORE)
  const int base = 1;
  HIP_DYNAMIC_SHARED(double, lds)
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  // nadv tracers are advected, tracer a of them is tracer (atr >> 8a) & 255 of the model (TKE and its length scale are
  // not unless use_TKEADV, mod_remap.F90:314-316 and every tracer loop after it)
  const int k = by_, ni = V.ni, nj = V.nj, ntr = V.ntr, t = threadIdx.x;
  constexpr int STX = FOLD ? RT_TW - 1 : RT_TW, STY = FOLD ? RT_TH - 1 : RT_TH;
  const int x0 = (bx_ % ntx) * STX, y0 = (bx_ / ntx) * STY;            // first point of the tile in the padded plane
  const bool zero_old = (tsel & 4) != 0;
  const bool lean = (tsel & 8) != 0;     // the mass, heat and salt fluxes go to uflx .. vsflx only: k_remap_update reads them there
  tsel &= 3;
  if (tsel) {
    // tsel 1: only the tiles that read no halo point (their 2-point rim lies inside 1..ii x 1..jj) -- they can run
    // while the halo exchange of cau, cav and the tracers is still under way; tsel 2: only the others
    // (a tile that holds the arctic seam: the exchange rewrites its last row, an interior one, phy/mod_xc.F90:4262-4372 -- tiles
    // whose rim reaches that row wait for it too)
    const int jlast = V.jj + NBDY - 1 - ((V.nreg == 2 && V.j0 + V.jj == V.jtdm) ? 1 : 0);
    const bool inner = x0 - 2 >= NBDY && x0 + RT_TW + 1 <= V.ii + NBDY - 1 && y0 - 2 >= NBDY && y0 + RT_TH + 1 <= jlast;
    if (inner != (tsel == 1)) return;
  }
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np, okm = (size_t)(k + mm) * np;
  // KPROF words of workgroup bx_ of level 10 (thread 0): 0 start, 1 phase 0 done, 2 gradients in LDS, 3 first pass done, 4 first round of the
  // batch loop done, 5 / 6 after the first / third batch, 7 end
  [[maybe_unused]] const bool kp_on = t == 0 && k == 10;
  if (kp_on) KPROF_MARK(bx_, 0);
  // LDS: [ scalars | ... ] then, from phase 2 on, [ gradient slots ] over the same space; cu, cv / corner velocities; masks
  double *const sc = lds;                                   // RT_NSC x RT_SN
  double *const gr = lds;                                   // RT_NG x RT_GN
  double *const cuv = lds + RT_NG(nadv) * RT_GN;            // 2 x RT_GN: cu, cv in phases 0-1, corner velocities after
  int *const mpl = (int *)(cuv + 2 * RT_GN);                // RT_SN masks
  double *const m2s = (double *)(mpl + RT_SN);              // 2 x RT_GN: scp2, scp2i of the gradient region (the faces' donor metrics)
  gci_t mpk = V.m[I_mpack];
  gcd_t scp2 = V.f[F_scp2], scp2i = V.f[F_scp2i];

  // ---- phase 0 -------------------------------------------------------------------------------------------------
  // (a) the scalar region: one point per thread
  double sv[4 + MAXTR];
  int sm;
  size_t cs_keep = 0;
  {
    gcd_t f_dp = V.f[F_dp] + okn, f_p = V.f[F_p] + (size_t)(k + 1) * np, f_t = V.f[F_temp] + okn;
    gcd_t f_s = V.f[F_saln] + okn, f_tr = V.f[F_trc] + okn;
    const int xs = x0 - 2 + t % RT_SW, ys = y0 - 2 + t / RT_SW;
    const bool in = t < RT_SN && xs >= 0 && xs < ni && ys >= 0 && ys < nj;
    const size_t cs = in ? (size_t)ys * ni + xs : 0;
    sm = in ? mpk[cs] : 0;
    sv[0] = f_dp[cs]; sv[1] = f_p[cs]; sv[2] = f_t[cs]; sv[3] = f_s[cs];
#pragma unroll
    for (int a = 0; a < MAXTR; a++) sv[4 + a] = a < nadv ? f_tr[cs + (size_t)L.idx[a] * 2 * V.kk * np] : 0.;
    cs_keep = cs;
  }
  // (b) this thread's point of the gradient region (threads 0 .. RT_GN-1): the tile's points first, then the rim
  int gx, gy;
  if (t < RT_TW * RT_TH) { gx = t % RT_TW + 1; gy = t / RT_TW + 1; }
  else {
    const int r = t - RT_TW * RT_TH;
    if (r < RT_GW) { gx = r; gy = 0; }
    else if (r < 2 * RT_GW) { gx = r - RT_GW; gy = RT_TH + 1; }
    else if (r < 2 * RT_GW + RT_TH) { gx = 0; gy = r - 2 * RT_GW + 1; }
    else { gx = RT_GW - 1; gy = r - 2 * RT_GW - RT_TH + 1; }
  }
  const bool gthread = t < RT_GN;
  const int q = gy * RT_GW + gx;                             // index in the gradient region
  const int sidx = (gy + 1) * RT_SW + gx + 1;                // index in the scalar region
  // Phase 0 is bound by the NUMBER of load instructions the workgroup issues (measured per workgroup: with a third of them gone it
  // shrank by a sixth; reordering them into one batch did nothing), so a point loads what only it can: cau, cav, scp2, scp2i and the
  // limiter's pressure at the point.  Its masks lie in the scalar region's copy (mpl), the metrics of its west and south neighbour
  // in the neighbours' copies (m2s): the face velocities cu, cv are formed in phase 1, from LDS.
  double cau_q, cav_q, pm, s2_q, s2i_q;
  bool gpoint;                                               // (completed in phase 1 with the point's own mask)
  {
    const int x = x0 - 1 + gx, y = y0 - 1 + gy;
    const bool inplane = gthread && x >= 0 && x < ni && y >= 0 && y < nj;
    const int i = x - (NBDY - 1), j = y - (NBDY - 1);
    // clamped into the plane: what a clamped point reads belongs to no point of any sweep's range
    const int xc = x < 1 ? 1 : (x > ni - 1 ? ni - 1 : x), yc = y < 1 ? 1 : (y > nj - 1 ? nj - 1 : y);
    const size_t c = (size_t)yc * ni + xc;
    cau_q = V.f[F_cau][c + ok];
    cav_q = V.f[F_cav][c + ok];
    s2_q = scp2[c];
    s2i_q = scp2i[c];
    pm = WK2(V, 0)[c];
    gpoint = inplane && j >= -1 && j <= V.jj + 2 && i >= -1 && i <= V.ii + 2;
  }
  // (c) this thread's face: threads 0..255 the u-face, threads 256..511 the v-face of tile point t mod 256.
  // Metrics of the face's donor cells, scp2 / scp2i at offset -1..1 ACROSS the face's normal and -1..0 ALONG it, are read in phase 3
  // from the gradient region's copy in LDS (m2s): as 12 loads of every face thread they were a third of the workgroup's load
  // instructions in phase 0 and 24 registers carried through the gradient phase.  (A face that is evaluated has 3 <= fx <= ni - 3,
  // likewise in y: its donors' region points are never the clamped ones.)
  const bool uface = t < RT_TW * RT_TH;
  const int ft = t % (RT_TW * RT_TH);
  const int fq = (ft / RT_TW + 1) * RT_GW + ft % RT_TW + 1;
  const int fx = x0 + ft % RT_TW, fy = y0 + ft / RT_TW;
  const bool fin = fx >= 1 && fx < ni - 1 && fy >= 1 && fy < nj - 1;      // every point with a face has neighbours
  const size_t fc = fin ? (size_t)fy * ni + fx : (size_t)ni + 1;
  const int fi = fx - (NBDY - 1), fj = fy - (NBDY - 1);
  const double caf = (uface ? V.f[F_cau] : V.f[F_cav])[fc + ok];
  const double pbf = (uface ? V.f[F_pbu] : V.f[F_pbv])[fc + (size_t)(n - 1) * np];
  gd_t const o_f = (uface ? V.f[F_uflx] : V.f[F_vflx]) + fc + okm;
  gd_t const o_ft = (uface ? V.f[F_utflx] : V.f[F_vtflx]) + fc + okm;
  gd_t const o_fs = (uface ? V.f[F_usflx] : V.f[F_vsflx]) + fc + okm;
  // old fluxes: the u-face accumulates (:1054-1056), the v-face assigns (:1455-1457).  Inside blomgpu_step init_fluxes has
  // zeroed them earlier in the step and nothing has added to them since (zero_old): 0 + flux without the read
  const bool own = !FOLD || (ft % RT_TW < RT_TW - 1 && ft / RT_TW < RT_TH - 1);
  const bool rd_old = uface && !zero_old && own;
  const double f_o = rd_old ? *o_f : 0., ft_o = rd_old ? *o_ft : 0., fs_o = rd_old ? *o_fs : 0.;

  if (t < RT_SN) {
    mpl[t] = sm;
#pragma unroll
    for (int s = 0; s < 4 + MAXTR; s++)
      if (s < 4 + nadv) sc[s * RT_SN + t] = sv[s];
  }
  if (gthread) { cuv[q] = cau_q; cuv[RT_GN + q] = cav_q; m2s[q] = s2_q; m2s[RT_GN + q] = s2i_q; }
  __syncthreads();
  if (kp_on) KPROF_MARK(bx_, 1);

  // ---- phase 1: gradients (k_remap_grad) and corner velocities (corner()) of this thread's point ------------------
  double gv[10 + 3 * MAXTR];
#pragma unroll
  for (int s = 0; s < 10 + 3 * MAXTR; s++) gv[s] = 0.;
  double cuc = 0., cvc = 0.;
  const int mpc = gthread ? mpl[sidx] : 0;                    // (0 outside the plane, as the scalar region stored it)
  gpoint = gpoint && MP(mpc);
  TNbr nb;
  nb.w = nb.e = nb.s = nb.n = nb.sw = nb.se = nb.nw = nb.ne = 0;
  double k_dxi = 0., k_dyi = 0., k_xd = 0., k_yd = 0.;       // kept for the tracers of later batches (MORE)
  if (gpoint) {
    // wet-restricted neighbours, mod_remap.F90:365-376 (wet_nbr of stage_advect.hip)
    const int a = MU(mpc), b = MU(mpl[sidx + 1]), d = MV(mpc), e = MV(mpl[sidx + RT_SW]);
    nb.w = sidx - a;
    nb.e = sidx + b;
    nb.s = sidx - d * RT_SW;
    nb.n = sidx + e * RT_SW;
    const int sw = sidx - a - d * RT_SW, se = sidx + b - d * RT_SW, nw = sidx - a + e * RT_SW, ne = sidx + b + e * RT_SW;
    nb.sw = MP(mpl[sw]) ? sw : sidx;
    nb.se = MP(mpl[se]) ? se : sidx;
    nb.nw = MP(mpl[nw]) ? nw : sidx;
    nb.ne = MP(mpl[ne]) ? ne : sidx;
    const int dxw = a + b, dyw = d + e;
    // 1 / max(1, ie - iw) and 1 / max(1, jn - js) (mod_remap.F90:377-378): the differences are 0, 1 or 2, the quotients 1 or exactly .5 --
    // selected, not divided (two of the gradient phase's eighteen fp64 divisions per point)
    const double dxi = dxw > 1 ? .5 : 1.;
    const double dyi = dyw > 1 ? .5 : 1.;
    const double *dp = sc, *plo = sc + RT_SN;
    // dp' = max(0,dp)+dpeps ; pup = plo - dp' ; lim = max(dpeps, min(pbmin - pup, dp'))
#define LIM(x) ({ const double d_ = hmax2(0., dp[x]) + DPEPS; hmax2(DPEPS, hmin2(pm - (plo[x] - d_), d_)); })
    const double dpsw = LIM(nb.sw), dps = LIM(nb.s), dpse = LIM(nb.se), dpw = LIM(nb.w), dpc = LIM(sidx);
    const double dpe = LIM(nb.e), dpnw = LIM(nb.nw), dpn = LIM(nb.n), dpne = LIM(nb.ne);
#undef LIM
    double dx = (dpe - dpw) * dxi, dy = (dpn - dps) * dyi;
    const double dgmx = .5 * (fabs(dx) + fabs(dy));
    const double dfmx = hmax2(0., hmax8(dpsw, dps, dpse, dpw, dpe, dpnw, dpn, dpne) - dpc);
    const double dfmn = hmin2(0., hmin8(dpsw, dps, dpse, dpw, dpe, dpnw, dpn, dpne) - dpc);
    const double dpt = hmax2(0., dp[sidx]) + DPEPS;
    double xd, yd;
    if (dfmx > 0. && dfmn < 0.) {
      const double qq = hmin2(dfmx / hmax2(dfmx, dgmx), dfmn / hmin2(dfmn, -dgmx));
      dx = dx * qq;
      dy = dy * qq;
      xd = dx / (12. * dpt);
      yd = dy / (12. * dpt);
    } else {
      dx = 0.; dy = 0.; xd = 0.; yd = 0.;
    }
    if (MORE) { k_dxi = dxi; k_dyi = dyi; k_xd = xd; k_yd = yd; }
    gv[G_DX] = dx;
    gv[G_DY] = dy;
    gv[8 + 3 * MAXTR] = dpt;                     // stored at RT_G_DPT / RT_G_PUP
    gv[9 + 3 * MAXTR] = plo[sidx] - dpt;
    limited_gradient_t(sc + 2 * RT_SN, nb, sidx, dxi, dyi, xd, yd, gv[G_TX], gv[G_TY], gv[G_TD]);
    limited_gradient_t(sc + 3 * RT_SN, nb, sidx, dxi, dyi, xd, yd, gv[G_SX], gv[G_SY], gv[G_SD]);
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++)
      if (nt < nadv)
        limited_gradient_t(sc + (4 + nt) * RT_SN, nb, sidx, dxi, dyi, xd, yd, gv[G_TRX(nt)], gv[G_TRY(nt)], gv[G_TRD(nt)]);
  }
  if (gthread && gx >= 1 && gy >= 1) {
    // corner velocities at the corner common to (x-1,y-1), (x,y-1), (x-1,y), (x,y), mod_remap.F90:623-659
    const int psw = MP(mpl[sidx - 1 - RT_SW]), pse = MP(mpl[sidx - RT_SW]), pnw = MP(mpl[sidx - 1]), pne = MP(mpc);
    const int nw = psw + pse + pnw + pne;
    // non-dimensional face velocities of the four faces that meet at the corner (mod_remap.F90:588-610): zero where no u- / v-point
    // exists; u-face of p: cau(p) scp2i(west of p) if it flows eastward, else cau(p) scp2i(p); v-face likewise with the south
    const double *const s2i = m2s + RT_GN, *const cav_r = cuv + RT_GN;
    auto cu_at = [&](int p, int mp) { const double ca = cuv[p], sw_ = s2i[p - 1], sc_ = s2i[p]; const double v_ = ca * (ca > 0. ? sw_ : sc_); return MU(mp) ? v_ : 0.; };
    auto cv_at = [&](int p, int mp) { const double ca = cav_r[p], ss_ = s2i[p - RT_GW], sc_ = s2i[p]; const double v_ = ca * (ca > 0. ? ss_ : sc_); return MV(mp) ? v_ : 0.; };
    const double cus = cu_at(q - RT_GW, mpl[sidx - RT_SW]), cun = cu_at(q, mpc);
    const double cvw = cv_at(q - 1, mpl[sidx - 1]), cve = cv_at(q, mpc);
    if (nw == 4) {
      cuc = (cus * cun <= 0.) ? 0. : 2. * cus * cun / (cus + cun);
      cvc = (cvw * cve <= 0.) ? 0. : 2. * cvw * cve / (cvw + cve);
    } else if (nw == 2) {
      if (psw + pse == 2) { cuc = cus; cvc = 0.; }
      else if (pnw + pne == 2) { cuc = cun; cvc = 0.; }
      else if (psw + pnw == 2) { cuc = 0.; cvc = cvw; }
      else if (pse + pne == 2) { cuc = 0.; cvc = cve; }
      else { cuc = 0.; cvc = 0.; }
    } else {
      cuc = 0.; cvc = 0.;
    }
  }
  __syncthreads();

  // ---- phase 2: the gradient fields replace the scalars ---------------------------------------------------------
  if (gthread) {
#pragma unroll
    for (int s = 0; s < 8 + 3 * MAXTR; s++)
      if (s < 8 + 3 * nadv) gr[s * RT_GN + q] = gv[s];
    gr[RT_G_DPT(nadv) * RT_GN + q] = gv[8 + 3 * MAXTR];
    gr[RT_G_PUP(nadv) * RT_GN + q] = gv[9 + 3 * MAXTR];
    cuv[q] = cuc;
    cuv[RT_GN + q] = cvc;
  }
  __syncthreads();
  if (kp_on) KPROF_MARK(bx_, 2);
  if (!MORE && !FOLD && !fin) return;

  // ---- phase 3: flux through this thread's face (k_remap_flux) ---------------------------------------------------------
  // (the face point's masks from the scalar region's copy in LDS: one load instruction of phase 0 less)
  const int mpf = fin ? mpl[(ft / RT_TW + 2) * RT_SW + ft % RT_TW + 2] : 0;
  const bool fmask = uface ? MU(mpf) : MV(mpf);
  const bool in_f = uface ? (fj >= 0 && fj <= V.jj + 1 && fi >= 0 && fi <= V.ii + 2) : (fj >= 0 && fj <= V.jj + 2 && fi >= 0 && fi <= V.ii + 1);
  if (!MORE && !FOLD && !in_f) return;
  const bool do_face = fin && in_f;
  // a polygon that does not exist (no triangle at that end; a face without flow) keeps weight zero: +-0 added to a sum that starts
  // at +0 never changes it, so the later batches add all three slots without a test (and the records stay in registers)
  PolyRec rec[3];
#pragma unroll
  for (int z = 0; z < 3; z++) { rec[z].fd = 0.; rec[z].qx = 0.; rec[z].qy = 0.; rec[z].x = fq; }
  Acc A;
  A.fd = 0.; A.ft = 0.; A.fs = 0.;
#pragma unroll
  for (int nt = 0; nt < MAXTR; nt++) A.ftr[nt] = 0.;
  if (do_face) {
  // cu resp. cv of the face (mod_remap.F90:588-610)
  const int f_along = uface ? 1 : RT_GW, f_across = uface ? RT_GW : 1;
  const double cf = fmask ? (caf > 0. ? caf * m2s[RT_GN + fq - f_along] : caf * m2s[RT_GN + fq]) : 0.;
  if (fmask) {
    const double cuc0 = cuv[fq], cvc0 = cuv[RT_GN + fq];
    const bool near = cf > 0.;                       // donor column i-1 (u-face) resp. donor row j-1 (v-face)
    const double sh = near ? .5 : -.5;
    const int fb = near ? fq - f_along : fq;         // the donor cell of the pentagon in the gradient region
    const double s2_c = m2s[fb], s2_m = m2s[fb - f_across], s2_p = m2s[fb + f_across];
    const double s2i_m = m2s[RT_GN + fb - f_across], s2i_p = m2s[RT_GN + fb + f_across];
    double a, ax, ay, axx, ayy, axy, x2, y2, x4, y4;
    if (uface) {
      const double cuc1 = cuv[fq + RT_GW], cvc1 = cuv[RT_GN + fq + RT_GW];
      const double ym = -.5 * (cvc0 + cvc1);
      const double xm = ((ym + .5) * cuc0 - (ym - .5) * cuc1 - 2. * cf) / (1. + cvc0 - cvc1);
      const int ic = near ? fq - 1 : fq;             // donor column
      if (cvc0 > 0.) {
        const double xc0 = (xm * cvc0 - cuc0 * (ym + .5)) / (cvc0 + ym + .5);
        const double xc1 = xc0 * s2_c * s2i_m;
        x4 = xc0 + sh;
        y4 = -.5;
        triint(s2_m, xc1 + sh, .5, -cuc0 + sh, -cvc0 + .5, sh, .5, a, ax, ay, axx, ayy, axy);
        add_contrib_t<MORE>(gr, nadv, ic - RT_GW, pbf, a, ax, ay, axx, ayy, axy, A, rec, 0);
      } else {
        x4 = -cuc0 + sh;
        y4 = -cvc0 - .5;
      }
      if (cvc1 < 0.) {
        const double xc0 = (xm * cvc1 - cuc1 * (ym - .5)) / (cvc1 + ym - .5);
        const double xc1 = xc0 * s2_c * s2i_p;
        x2 = xc0 + sh;
        y2 = .5;
        triint(s2_p, xc1 + sh, -.5, sh, -.5, -cuc1 + sh, -cvc1 - .5, a, ax, ay, axx, ayy, axy);
        add_contrib_t<MORE>(gr, nadv, ic + RT_GW, pbf, a, ax, ay, axx, ayy, axy, A, rec, 1);
      } else {
        x2 = -cuc1 + sh;
        y2 = -cvc1 + .5;
      }
      penint(s2_c, sh, .5, x2, y2, xm + sh, ym, x4, y4, sh, -.5, a, ax, ay, axx, ayy, axy);
      add_contrib_t<MORE>(gr, nadv, ic, pbf, a, ax, ay, axx, ayy, axy, A, rec, 2);
      // mod_remap.F90:1054-1056
      if (base && own) {
        *o_f = f_o + A.fd;
        *o_ft = ft_o + A.ft;
        *o_fs = fs_o + A.fs;
      }
    } else {
      const double cuc1 = cuv[fq + 1], cvc1 = cuv[RT_GN + fq + 1];
      const double xm = -.5 * (cuc0 + cuc1);
      const double ym = ((xm + .5) * cvc0 - (xm - .5) * cvc1 - 2. * cf) / (1. + cuc0 - cuc1);
      const int jc = near ? fq - RT_GW : fq;         // donor row
      if (cuc0 > 0.) {
        const double yc0 = (ym * cuc0 - cvc0 * (xm + .5)) / (cuc0 + xm + .5);
        const double yc1 = yc0 * s2_c * s2i_m;
        x2 = -.5;
        y2 = yc0 + sh;
        triint(s2_m, .5, yc1 + sh, .5, sh, -cuc0 + .5, -cvc0 + sh, a, ax, ay, axx, ayy, axy);
        add_contrib_t<MORE>(gr, nadv, jc - 1, pbf, a, ax, ay, axx, ayy, axy, A, rec, 0);
      } else {
        x2 = -cuc0 - .5;
        y2 = -cvc0 + sh;
      }
      if (cuc1 < 0.) {
        const double yc0 = (ym * cuc1 - cvc1 * (xm - .5)) / (cuc1 + xm - .5);
        const double yc1 = yc0 * s2_c * s2i_p;
        x4 = .5;
        y4 = yc0 + sh;
        triint(s2_p, -.5, yc1 + sh, -cuc1 - .5, -cvc1 + sh, -.5, sh, a, ax, ay, axx, ayy, axy);
        add_contrib_t<MORE>(gr, nadv, jc + 1, pbf, a, ax, ay, axx, ayy, axy, A, rec, 1);
      } else {
        x4 = -cuc1 + .5;
        y4 = -cvc1 + sh;
      }
      penint(s2_c, -.5, sh, x2, y2, xm, ym + sh, x4, y4, .5, sh, a, ax, ay, axx, ayy, axy);
      add_contrib_t<MORE>(gr, nadv, jc, pbf, a, ax, ay, axx, ayy, axy, A, rec, 2);
      // mod_remap.F90:1455-1457: assignment (not accumulation) for the v-components
      if (base && own) {
        *o_f = A.fd;
        *o_ft = A.ft;
        *o_fs = A.fs;
      }
    }
  }
  // the flux planes of k_remap_update: W_FDU, W_FDV, W_FTU, W_FTV, .. alternate
  const int off = uface ? 0 : 1;
  if (!FOLD) {
    if (base && !lean) {
      WK(V, W_FDU(ntr) + off)[fc + ok] = A.fd;
      WK(V, W_FTU(ntr) + off)[fc + ok] = A.ft;
      WK(V, W_FSU(ntr) + off)[fc + ok] = A.fs;
    }
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++)
      if (nt < nadv) WK(V, W_FTRU(ntr, L.idx[nt]) + off)[fc + ok] = A.ftr[nt];
  }
  }
  // ---- FOLD: the update of the owned cells, mod_remap.F90:1468-1520 ------------------------------------------------------
  // thread t < 256 is the cell whose west face it evaluated; its east face is thread t + 1's, its south face thread 256 + t's,
  // its north face thread 256 + t + RT_TW's: the fluxes meet in LDS (the gradient slots have been read by now)
  const int cx = fx - (NBDY - 1), cy = fy - (NBDY - 1);
  const bool upd = FOLD && uface && own && fin && cy >= 0 && cy <= V.jj + 1 && cx >= 0 && cx <= V.ii + 1 && MP(mpf);
  double q_dp = 0., dpn = 1.;
  double *const fl = lds;
  if (FOLD) {
    __syncthreads();
    fl[t] = A.fd; fl[RT_NT + t] = A.ft; fl[2 * RT_NT + t] = A.fs;
#pragma unroll
    for (int nt = 0; nt < MAXTR; nt++)
      if (nt < nadv) fl[(3 + nt) * RT_NT + t] = A.ftr[nt];
    __syncthreads();
    if (upd) {
      const int te = t + 1, ts = RT_TW * RT_TH + t, tn = ts + RT_TW;
      const double s2i = m2s[RT_GN + fq];
      q_dp = fmax2(0., V.f[F_dp][fc + okn]) + DPEPS;
      dpn = q_dp - (fl[te] - fl[t] + fl[tn] - fl[ts]) * s2i;
      const double told = V.f[F_temp][fc + okn], sold = V.f[F_saln][fc + okn];
      WK(V, R_T(ntr))[fc + ok] = (q_dp * told - (fl[RT_NT + te] - fl[RT_NT + t] + fl[RT_NT + tn] - fl[RT_NT + ts]) * s2i) / dpn;
      WK(V, R_S(ntr))[fc + ok] = (q_dp * sold - (fl[2 * RT_NT + te] - fl[2 * RT_NT + t] + fl[2 * RT_NT + tn] - fl[2 * RT_NT + ts]) * s2i) / dpn;
#pragma unroll
      for (int nt = 0; nt < MAXTR; nt++)
        if (nt < nadv) {
          const double *f = fl + (3 + nt) * RT_NT;
          const double xold = V.f[F_trc][fc + okn + (size_t)L.idx[nt] * 2 * V.kk * np];
          WK(V, R_TR(ntr, L.idx[nt]))[fc + ok] = (q_dp * xold - (f[te] - f[t] + f[tn] - f[ts]) * s2i) / dpn;
        }
      WK(V, R_DP(ntr))[fc + ok] = fmax2(0., dpn - DPEPS);
    }
  }
  if (!MORE) {
    if (kp_on) KPROF_MARK(bx_, 3);
    return;
  }
  // ---- the other advected tracers, a batch at a time --------------------------------------------------------------------------
  gcd_t f_tr = V.f[F_trc] + okn;
  const int off2 = uface ? 0 : 1;
  if constexpr (!FOLD) {
    // A batch of RT_NB tracers in LDS: its scalars S (RT_NB x RT_SN), behind them its limited gradients G (3 RT_NB x RT_GN).
    // Two barriers per batch, and the work between them comes from two batches:
    //   phase 1  G <- the gradients of this batch (in registers since the previous phase 2), S <- the scalars of the next one
    //            (in registers since the phase 2 before that)
    //   phase 2  the loads of the scalars of the batch after the next are issued; the fluxes of this batch from G and the
    //            polygons' records; the gradients of the next batch from S into registers
    // so a phase 2 holds global loads, LDS reads of two kinds and two independent strands of arithmetic.
    double *const bs = lds, *const bg = lds + RT_NB * RT_SN;
    double tv[RT_NB], gb[3 * RT_NB];
#pragma unroll
    for (int a = 0; a < 3 * RT_NB; a++) gb[a] = 0.;
    auto fetch = [&](int b0) {
#pragma unroll
      for (int a = 0; a < RT_NB; a++)
        tv[a] = (b0 + a < nadv_all && t < RT_SN) ? f_tr[cs_keep + (size_t)L.idx[b0 + a < 64 ? b0 + a : 0] * 2 * V.kk * np] : 0.;
    };
    fetch(nadv);
    __syncthreads();                                           // the gradient slots of the first pass have been read
    if (kp_on) KPROF_MARK(bx_, 3);
    for (int b0 = nadv - RT_NB; b0 < nadv_all; b0 += RT_NB) {  // (the first round only fills S and evaluates the first gradients)
      const bool cur = b0 >= nadv, next = b0 + RT_NB < nadv_all;
      const int nb_ = nadv_all - b0 < RT_NB ? nadv_all - b0 : RT_NB;
      if (cur && gthread) {
#pragma unroll
        for (int a = 0; a < 3 * RT_NB; a++) bg[a * RT_GN + q] = gb[a];
      }
      if (next && t < RT_SN) {
#pragma unroll
        for (int a = 0; a < RT_NB; a++) bs[a * RT_SN + t] = tv[a];
      }
      __syncthreads();
      if (next) fetch(b0 + 2 * RT_NB);
      if (cur && do_face) {
#pragma unroll
        for (int a = 0; a < RT_NB; a++)
          if (a < nb_) {
            double f = 0.;
#pragma unroll
            for (int pz = 0; pz < 3; pz++)
              f = f + rec[pz].fd * bg[(3 * a + 2) * RT_GN + rec[pz].x] + rec[pz].qx * bg[(3 * a) * RT_GN + rec[pz].x] +
                  rec[pz].qy * bg[(3 * a + 1) * RT_GN + rec[pz].x];
            WK(V, W_FTRU(ntr, L.idx[b0 + a]) + off2)[fc + ok] = f;
          }
      }
      if (next && gthread) {
        const int nn_ = nadv_all - (b0 + RT_NB) < RT_NB ? nadv_all - (b0 + RT_NB) : RT_NB;
#pragma unroll
        for (int a = 0; a < RT_NB; a++)
          if (a < nn_) {
            double gxx = 0., gyy = 0., gdd = 0.;
            if (gpoint) limited_gradient_t(bs + a * RT_SN, nb, sidx, k_dxi, k_dyi, k_xd, k_yd, gxx, gyy, gdd);
            gb[3 * a] = gxx; gb[3 * a + 1] = gyy; gb[3 * a + 2] = gdd;
          }
      }
      __syncthreads();
      if (kp_on) {
        const int it_ = (b0 - (nadv - RT_NB)) / RT_NB;
        if (it_ == 0) KPROF_MARK(bx_, 4);
        else if (it_ == 1) KPROF_MARK(bx_, 5);
        else if (it_ == 3) KPROF_MARK(bx_, 6);
      }
    }
    if (kp_on) KPROF_MARK(bx_, 7);
    return;
  }
  // (with the update folded in: the batches one after the other, three barriers each)
  // the scalars of a batch are loaded while the batch before it is worked on: tv is free again once it has gone to LDS
  double tv[MAXTR];
#pragma unroll
  for (int a = 0; a < MAXTR; a++) tv[a] = (nadv + a < nadv_all && t < RT_SN) ? f_tr[cs_keep + (size_t)L.idx[nadv + a] * 2 * V.kk * np] : 0.;
  for (int b0 = nadv; b0 < nadv_all; b0 += MAXTR) {
    const int nb_ = nadv_all - b0 < MAXTR ? nadv_all - b0 : MAXTR;
    __syncthreads();                                         // the gradient slots of the previous batch have been read
    if (t < RT_SN) {
#pragma unroll
      for (int a = 0; a < MAXTR; a++)
        if (a < nb_) sc[a * RT_SN + t] = tv[a];
    }
#pragma unroll
    for (int a = 0; a < MAXTR; a++)
      tv[a] = (b0 + MAXTR + a < nadv_all && t < RT_SN) ? f_tr[cs_keep + (size_t)L.idx[b0 + MAXTR + a] * 2 * V.kk * np] : 0.;
    __syncthreads();
    // the scalars lie in slots 0..3 of the scalar region (below RT_SN * MAXTR doubles), the tracer gradient slots start at
    // 8 * RT_GN: no overlap, the gradients go straight to their slots
    static_assert(MAXTR * RT_SN <= 8 * RT_GN, "batch scalars and tracer gradient slots must not overlap");
    if (gthread) {
#pragma unroll
      for (int a = 0; a < MAXTR; a++)
        if (a < nb_) {
          double gxx = 0., gyy = 0., gdd = 0.;
          if (gpoint) limited_gradient_t(sc + a * RT_SN, nb, sidx, k_dxi, k_dyi, k_xd, k_yd, gxx, gyy, gdd);
          gr[G_TRX(a) * RT_GN + q] = gxx;
          gr[G_TRY(a) * RT_GN + q] = gyy;
          gr[G_TRD(a) * RT_GN + q] = gdd;
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < MAXTR; a++)
      if (a < nb_) fl[a * RT_NT + t] = 0.;
    // (the batch's scalars, which lay where the fluxes now are, were last read before the barrier above; a face that
    // is not evaluated contributes nothing)
    if (do_face) {
#pragma unroll
      for (int a = 0; a < MAXTR; a++)
        if (a < nb_) {
          double f = 0.;
#pragma unroll
          for (int pz = 0; pz < 3; pz++)
            f = f + rec[pz].fd * gr[G_TRD(a) * RT_GN + rec[pz].x] + rec[pz].qx * gr[G_TRX(a) * RT_GN + rec[pz].x] +
                rec[pz].qy * gr[G_TRY(a) * RT_GN + rec[pz].x];
          fl[a * RT_NT + t] = f;
        }
    }
    __syncthreads();
    if (upd) {
      const int te = t + 1, ts = RT_TW * RT_TH + t, tn = ts + RT_TW;
      const double s2i = m2s[RT_GN + fq];
#pragma unroll
      for (int a = 0; a < MAXTR; a++)
        if (a < nb_) {
          const double *f = fl + a * RT_NT;
          const double xold = V.f[F_trc][fc + okn + (size_t)L.idx[b0 + a] * 2 * V.kk * np];
          WK(V, R_TR(ntr, L.idx[b0 + a]))[fc + ok] = (q_dp * xold - (f[te] - f[t] + f[tn] - f[ts]) * s2i) / dpn;
        }
    }
  }
}

// FOLD: what k_remap_update leaves in the HALO of the fields.  It updates the cells 0..ii+1 x 0..jj+1, one ring beyond the tile,
// and commits dp = max(0, dp) + dpeps on the two rings beyond that (mod_remap.F90:297-303); the tile kernel cannot write halo
// cells in place (other workgroups read them in their rims), and pbcor1 takes the interior from the work space.  Most of
// these halo values are overwritten by the next halo update before anything reads them, but not all arrays get one before
// the step ends (tracers that are advected but not diffused), and the state after a step is the reference's to the halo.
__global__ void k_remap_ring(const DevView *__restrict__ Vp, int nn, int nadv_all, AdvList L) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj, w = ii + 6;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int i, j;
  if (t < 6 * w) {                                     // rows -2..0 and jj+1..jj+3, i = -2..ii+3
    const int r = t / w;
    j = r < 3 ? r - 2 : jj + 1 + (r - 3);
    i = t % w - 2;
  } else {
    t -= 6 * w;
    if (t >= 6 * jj) return;                            // rows 1..jj: i = -2..0 and ii+1..ii+3
    j = t / 6 + 1;
    const int q = t % 6;
    i = q < 3 ? q - 2 : ii + 1 + (q - 3);
  }
  const size_t c = IDX(V, i, j);
  if (!V.m[I_ip][c]) return;
  const int k = blockIdx.y, ntr = V.ntr;
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np;
  if (j < 0 || j > jj + 1 || i < 0 || i > ii + 1) {
    V.f[F_dp][c + okn] = fmax2(0., V.f[F_dp][c + okn]) + DPEPS;
    return;
  }
  V.f[F_dp][c + okn] = WK(V, R_DP(ntr))[c + ok];
  V.f[F_temp][c + okn] = WK(V, R_T(ntr))[c + ok];
  V.f[F_saln][c + okn] = WK(V, R_S(ntr))[c + ok];
  for (int a = 0; a < nadv_all; a++)
    V.f[F_trc][c + okn + (size_t)L.idx[a] * 2 * V.kk * np] = WK(V, R_TR(ntr, L.idx[a]))[c + ok];
}

int remap_tile_launch(blomgpu_ctx *c, int n, int mm, int nn, int tsel, bool fold) {
  const bool zeroed = c->in_sequence && c->fluxes_zeroed;
  if (tsel == 0 || tsel == 2) c->fluxes_zeroed = false;             // (a split launch: tiles 1, then tiles 2)
  if (zeroed) tsel |= 4;
  if (zeroed && !fold && c->lean_fluxes) tsel |= 8;
  // the lean init_fluxes left last step's fluxes in every plane this launch does not STORE to: any other path would add to them
  if (c->fluxes_lean && !(tsel & 8)) return ctx_fail(c, "remap: the flux arrays were zeroed on their ring only (lean init_fluxes) but the storing tile path is not taken");
  if (tsel == (tsel | 8) && ((tsel & 3) == 0 || (tsel & 3) == 2)) c->fluxes_lean = false;
  const DevView &h = c->h;
  const int stx = fold ? RT_TW - 1 : RT_TW, sty = fold ? RT_TH - 1 : RT_TH;
  const int ntx = (h.ni + stx - 1) / stx, nty = (h.nj + sty - 1) / sty;
  static_assert(RT_NSC(MAXTR) * RT_SN <= RT_NG(MAXTR) * RT_GN, "the scalars must fit under the gradient slots");
  if (h.ntr > 255) return ctx_fail(c, "remap: tracer indices are packed in 8 bits");
  if (h.ntr > 0 && W_FTRV(h.ntr, h.ntr - 1) >= h.nwk) return ctx_fail(c, "remap: work space too small for this many tracers");
  // the advected tracers (TKE and its length scale are not advected unless use_TKEADV, mod_remap.F90:314-316)
  AdvList L;
  int nadv = 0;
  for (int nt = 0; nt < h.ntr; nt++)
    if (!trc_skip_adv(h.P, nt + 1)) {
      if (nadv == 64) return ctx_fail(c, "remap: more than 64 advected tracers");
      L.idx[nadv++] = nt;
    }
  for (int a = nadv; a < 64; a++) L.idx[a] = 0;
  // without the fold, how many of more than MAXTR tracers ride with the first pass is an option (remap_nfirst): with none the first
  // pass needs 34 KB of LDS and a batch 46 KB instead of 67 KB, but at 128 VGPRs two workgroups per CU is the limit anyway and
  // the first pass's tracers are cheaper than a batch's: 3.57 ms with four against 3.74 ms with none (24 tracers, tnx1v4s)
  int nfirst = nadv < MAXTR ? nadv : MAXTR;
  if (nadv > MAXTR && !fold) nfirst = c->remap_nfirst < MAXTR ? (c->remap_nfirst > 0 ? c->remap_nfirst : 0) : MAXTR;
  size_t lds = sizeof(double) * (RT_NG(nfirst) * RT_GN + 2 * RT_GN) + sizeof(int) * RT_SN + sizeof(double) * 2 * RT_GN;
  if (nadv > MAXTR && !fold && lds < sizeof(double) * RT_NB * (RT_SN + 3 * RT_GN)) lds = sizeof(double) * RT_NB * (RT_SN + 3 * RT_GN);
  const dim3 grid(ntx * nty, h.kk);
  if (fold) {
    {
      TimeScope tk(c, "k_remap_tile");
      if (nadv > MAXTR) hipLaunchKernelGGL((k_remap_tile<true, true>), grid, dim3(RT_NT), lds, c->stream, c->d, n, mm, nn, ntx, nadv, nfirst, L, tsel KPROF_PASS(9));
      else hipLaunchKernelGGL((k_remap_tile<false, true>), grid, dim3(RT_NT), lds, c->stream, c->d, n, mm, nn, ntx, nadv, nfirst, L, tsel KPROF_PASS(9));
    }
    if ((tsel & 3) != 1)                                // (a split launch: after the second part)
      hipLaunchKernelGGL(k_remap_ring, dim3((6 * (h.ii + 6) + 6 * h.jj + 63) / 64, h.kk), dim3(64), 0, c->stream, c->d, nn, nadv, L);
  } else {
    TimeScope tk(c, "k_remap_tile");
    if (nadv > MAXTR) hipLaunchKernelGGL((k_remap_tile<true, false>), grid, dim3(RT_NT), lds, c->stream, c->d, n, mm, nn, ntx, nadv, nfirst, L, tsel KPROF_PASS(9));
    else hipLaunchKernelGGL((k_remap_tile<false, false>), grid, dim3(RT_NT), lds, c->stream, c->d, n, mm, nn, ntx, nadv, nfirst, L, tsel KPROF_PASS(9));
  }
  return 0;
}
