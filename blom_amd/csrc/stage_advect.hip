// advect -- flux areas + incremental remapping of layer thickness, T, S and tracers.
// phy/mod_advect.F90:59-189 (advmth='remap') and phy/mod_remap.F90:53-1522.
//
// The reference calls remap() once per layer on 2-D slices; layers are independent, so each
// kernel here covers all layers at once (blockIdx.y = layer) with one thread per point of the
// padded plane (unit-stride in i across the wavefront):
//   k_adv_flux_area  cau,cav = clamp(u*dt*scuy + ...)                        (mod_advect:71-94)
//   k_adv_pbmin      9-point wet-aware minimum of bottom pressure            (mod_advect:100-121)
//   k_remap_tile     limited gradients, corner velocities, u-face and v-face flux polygon integrals (<= 2 triangles +
//                    1 pentagon each), accumulation into uflx..: one LDS-tiled kernel (stage_remap_tile.hip; mod_remap:358-1462)
//   k_remap_update   flux-divergence update of dp,T,S,trc                     (mod_remap:1468-1520)
// "dp = max(0,dp)+dpeps; pup = plo-dp" (mod_remap:297-303) is applied on the fly wherever a
// cell is read (every donor/neighbour cell is wet and inside the reference's -2..+3 range), and
// committed for the outer ring by k_remap_update, which reproduces the reference's in-place
// side effect on halo cells.
// The four directional variants of each face in the Fortran differ only by the donor cell and
// the sign of the half-cell offset; they are folded into one body with sh = +/-0.5 (exact).
//
// Algorithmic bytes: (25 + 2*ntr) F; the flux planes cross HBM once between the tile kernel and the
// update (the update cannot be written in place: the neighbouring tiles read the old values in their rims).
// Roofline: HBM.
#include "remap_common.h"

#define THREAD_IJ(V)                                                       \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// ---- mod_advect.F90:71-94 --------------------------------------------------------------------
__global__ void k_adv_flux_area(const DevView *__restrict__ Vp, int m, int mm, int nn) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const int k = by_;
  const size_t np = V.nplane, okm = c + (size_t)(k + mm) * np, okn = c + (size_t)(k + nn) * np;
  const size_t om = c + (size_t)(m - 1) * np;
  const double delt1 = V.P.delt1, dlt = V.P.dlt;
  if (V.m[I_iu][c]) {
    const double dtdl = delt1 * V.f[F_scuy][c];
    const double ca_tmp = V.f[F_u][okm] * dtdl + V.f[F_ubflxs_p][om] * dlt / V.f[F_pbu][om] +
                          (V.f[F_umfltd][okm] + V.f[F_umflsm][okm]) / fmax2(ONEMM, V.f[F_dpu][okn]);
    const double um = V.f[F_umax][c];
    V.f[F_cau][c + (size_t)k * np] = fmax2(-um * dtdl, fmin2(um * dtdl, ca_tmp));
  }
  if (V.m[I_iv][c]) {
    const double dtdl = delt1 * V.f[F_scvx][c];
    const double ca_tmp = V.f[F_v][okm] * dtdl + V.f[F_vbflxs_p][om] * dlt / V.f[F_pbv][om] +
                          (V.f[F_vmfltd][okm] + V.f[F_vmflsm][okm]) / fmax2(ONEMM, V.f[F_dpv][okn]);
    const double vm = V.f[F_vmax][c];
    V.f[F_cav][c + (size_t)k * np] = fmax2(-vm * dtdl, fmin2(vm * dtdl, ca_tmp));
  }
}

// wet-restricted neighbour plane offsets, mod_remap.F90:365-376 == mod_advect.F90:103-114
struct Nbr {
  size_t w, e, s, n, sw, se, nw, ne;
  int dxw, dyw;   // ie-iw, jn-js
};
__device__ inline Nbr wet_nbr(const DevView &V, size_t c) {
  gci_t ip = V.m[I_ip], iu = V.m[I_iu], iv = V.m[I_iv];
  const int ni = V.ni;
  const int a = iu[c], b = iu[c + 1], d = iv[c], e = iv[c + ni];
  Nbr r;
  r.w = c - a;
  r.e = c + b;
  r.s = c - (size_t)d * ni;
  r.n = c + (size_t)e * ni;
  r.dxw = a + b;
  r.dyw = d + e;
  // corner (iw,js): if wet use it, else fall back to the centre cell
  const size_t sw = c - a - (size_t)d * ni, se = c + b - (size_t)d * ni;
  const size_t nw = c - a + (size_t)e * ni, ne = c + b + (size_t)e * ni;
  r.sw = ip[sw] ? sw : c;
  r.se = ip[se] ? se : c;
  r.nw = ip[nw] ? nw : c;
  r.ne = ip[ne] ? ne : c;
  return r;
}

// ---- mod_advect.F90:100-121 ---------------------------------------------------------------------
__global__ void k_adv_pbmin(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2 || !V.m[I_ip][c]) return;
  gcd_t pb = V.f[F_p] + (size_t)V.kk * V.nplane;
  const Nbr b = wet_nbr(V, c);
  double r = fmin2(pb[b.sw], pb[b.s]);
  r = fmin2(r, pb[b.se]);
  r = fmin2(r, pb[b.w]);
  r = fmin2(r, pb[c]);
  r = fmin2(r, pb[b.e]);
  r = fmin2(r, pb[b.nw]);
  r = fmin2(r, pb[b.n]);
  r = fmin2(r, pb[b.ne]);
  WK2(V, 0)[c] = r;
}

// ---- mod_remap.F90:1468-1520 (+ the in-place dp side effect of :297-303 on the outer ring) --------
// mmlean >= 0 (inside blomgpu_step, where the flux arrays of level m hold nothing but this call's fluxes -- init_fluxes zeroed them
// earlier in the step): the mass, heat and salt fluxes of the faces are read from uflx .. vsflx (k + mmlean) themselves, where the tile
// kernel has just stored them, instead of from six more work planes holding the same numbers (a face without a velocity point
// carries no flux; 0 + f and f may differ in the sign of a zero, which no result of this update can see: q > 0, and a zero
// difference of fluxes is subtracted from q dp, q T, q S)
__global__ void k_remap_update(const DevView *__restrict__ Vp, int nn, int mmlean) {
  const DevView &V = *Vp;
  THREAD_IJ(V);
  if (j < -2 || j > V.jj + 3 || i < -2 || i > V.ii + 3 || !V.m[I_ip][c]) return;
  const int k = by_, ntr = V.ntr;
  const size_t np = V.nplane, ok = (size_t)k * np, okn = (size_t)(k + nn) * np;
  gd_t dp = V.f[F_dp] + okn;
  const double q = fmax2(0., dp[c]) + DPEPS;
  if (j < 0 || j > V.jj + 1 || i < 0 || i > V.ii + 1) {
    dp[c] = q;
    return;
  }
  const size_t e = c + 1, nb = c + V.ni;
  const double s2i = V.f[F_scp2i][c];
  double dfd, dft, dfs;
  if (mmlean >= 0) {
    const size_t okm = (size_t)(k + mmlean) * np;
    gci_t mpk = V.m[I_mpack];
    const bool uc = (mpk[c] >> 1) & 1, ue = (mpk[e] >> 1) & 1, vc = (mpk[c] >> 2) & 1, vn = (mpk[nb] >> 2) & 1;
    gcd_t fdu = V.f[F_uflx] + okm, fdv = V.f[F_vflx] + okm, ftu = V.f[F_utflx] + okm, ftv = V.f[F_vtflx] + okm;
    gcd_t fsu = V.f[F_usflx] + okm, fsv = V.f[F_vsflx] + okm;
    dfd = (ue ? fdu[e] : 0.) - (uc ? fdu[c] : 0.) + (vn ? fdv[nb] : 0.) - (vc ? fdv[c] : 0.);
    dft = (ue ? ftu[e] : 0.) - (uc ? ftu[c] : 0.) + (vn ? ftv[nb] : 0.) - (vc ? ftv[c] : 0.);
    dfs = (ue ? fsu[e] : 0.) - (uc ? fsu[c] : 0.) + (vn ? fsv[nb] : 0.) - (vc ? fsv[c] : 0.);
  } else {
    gcd_t fdu = WK(V, W_FDU(ntr)) + ok, fdv = WK(V, W_FDV(ntr)) + ok;
    gcd_t ftu = WK(V, W_FTU(ntr)) + ok, ftv = WK(V, W_FTV(ntr)) + ok;
    gcd_t fsu = WK(V, W_FSU(ntr)) + ok, fsv = WK(V, W_FSV(ntr)) + ok;
    dfd = fdu[e] - fdu[c] + fdv[nb] - fdv[c];
    dft = ftu[e] - ftu[c] + ftv[nb] - ftv[c];
    dfs = fsu[e] - fsu[c] + fsv[nb] - fsv[c];
  }
  const double dpn = q - dfd * s2i;
  gd_t temp = V.f[F_temp] + okn, saln = V.f[F_saln] + okn;
  temp[c] = (q * temp[c] - dft * s2i) / dpn;
  saln[c] = (q * saln[c] - dfs * s2i) / dpn;
  for (int nt = 0; nt < ntr; nt++) {
    if (trc_skip_adv(V.P, nt + 1)) continue;                   // phy/mod_remap.F90:1497-1499
    gd_t tr = V.f[F_trc] + okn + (size_t)nt * 2 * V.kk * np;
    gcd_t fu = WK(V, W_FTRU(ntr, nt)) + ok, fv = WK(V, W_FTRV(ntr, nt)) + ok;
    tr[c] = (q * tr[c] - (fu[e] - fu[c] + fv[nb] - fv[c]) * s2i) / dpn;
  }
  dp[c] = fmax2(0., dpn - DPEPS);
}

int st_advect(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  const DevView &h = c->h;
  const size_t np = h.nplane;
  hipLaunchKernelGGL(k_adv_flux_area, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, m, mm, nn);
  if (h.P.advmth == 1) {                                                              // mod_advect:155-164
    {
      TimeScope ts(c, "cppm");
      if (int rc = st_cppm(c, m, n, mm, nn, k1m, k1n)) return rc;
    }
    if (int rc = st_xctilr(c, h.f[F_dp] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    if (int rc = st_xctilr(c, h.f[F_temp] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    if (int rc = st_xctilr(c, h.f[F_saln] + (size_t)(k1n - 1) * np, 1, h.kk, 1, 1, 1)) return rc;
    for (int nt = 0; nt < h.ntr; nt++)
      if (int rc = st_xctilr(c, h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np, 1, h.kk, 1, 1, 1)) return rc;
    return 0;
  }
  hipLaunchKernelGGL(k_adv_pbmin, plane_grid(h), dim3(256), 0, c->stream, c->d);
  std::vector<double *> ptrs = {h.f[F_cau], h.f[F_cav]};                              // mod_advect:124-131
  std::vector<int> nl = {h.kk, h.kk}, it = {13, 14};
  for (int nt = 0; nt < h.ntr; nt++) {
    if (trc_skip_adv(h.P, nt + 1)) continue;                                          // :127-129
    ptrs.push_back(h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np);
    nl.push_back(h.kk);
    it.push_back(1);
  }
  const int nf = (int)ptrs.size();
  // st_xctilr_multi takes up to 16 stacks per launch
  auto halo_all = [&]() -> int {
    for (int f = 0; f < nf; f += 16)
      if (int rc = st_xctilr_multi(c, nf - f < 16 ? nf - f : 16, ptrs.data() + f, nl.data() + f, 3, 3, it.data() + f)) return rc;
    return 0;
  };
  // RCCL tiles: the exchange (pack, send/recv, unpack) goes to the second stream and the tiles of k_remap_tile whose
  // rim lies inside the tile -- they read no halo point of cau, cav or the tracers -- run meanwhile; the tiles along the
  // edge follow when the halos have landed.  The exchange reads interior strips and writes halo points only.
  // inside blomgpu_step the tile kernel also does the update and hands the new dp, T, S, tracers to pbcor1 through the work space
  const bool fold = c->in_sequence && c->remap_fold;
  // the flux arrays of level m were zeroed in this step and nothing has added to them: the tile kernel's fluxes ARE uflx .. vsflx (m)
  const int mmlean = !fold && c->in_sequence && c->fluxes_zeroed && c->lean_fluxes ? mm : -1;
  const bool ovl = c->tiling.rccl && c->xstream && c->halo_overlap && !c->timing;
  if (ovl) {
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->xstream, c->ev_fork, 0));
    c->halo_stream = c->xstream;
    const int rc = halo_all();
    c->halo_stream = nullptr;
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_join, c->xstream));
    if (int rc2 = remap_tile_launch(c, n, mm, nn, 1, fold)) return rc2;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    if (int rc2 = remap_tile_launch(c, n, mm, nn, 2, fold)) return rc2;
    if (!fold) hipLaunchKernelGGL(k_remap_update, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn, mmlean);
  } else {
    if (int rc = halo_all()) return rc;
    TimeScope ts(c, "remap");
    if (int rc = remap_tile_launch(c, n, mm, nn, 0, fold)) return rc;
    if (!fold) hipLaunchKernelGGL(k_remap_update, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn, mmlean);
  }
  c->remap_handed_over = fold;
  HIPCHK(c, hipGetLastError());
  return 0;
}
