// init_fluxes, tmsmt1, tmsmt2, initms and the shared "interface pressure + velocity-point
// thickness" kernels (tail of tmsmt2 / mxlayr, head of pgforc).
//
// All of these are pure streaming: one thread per point of the padded (ni x nj) plane,
// consecutive lanes on consecutive i (unit stride, 8 B/lane), layers on blockIdx.y or in a
// per-column loop where a k-recurrence exists.  Roofline: HBM.
#include "blomgpu_internal.h"

#define PLANE_IJ(V)                                                        \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_;                                                     \
  (void)i; (void)j; (void)c

// ---- init_fluxes, phy/mod_state.F90:352-372 ------------------------------------------------
__global__ void k_init_fluxes(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 0 || j > V.jj + 2 || i < 0 || i > V.ii + 2) return;
  const size_t o = c + (size_t)(by_ + mm) * V.nplane;
  if (V.m[I_iu][c]) { V.f[F_uflx][o] = 0.; V.f[F_utflx][o] = 0.; V.f[F_usflx][o] = 0.; }
  if (V.m[I_iv][c]) { V.f[F_vflx][o] = 0.; V.f[F_vtflx][o] = 0.; V.f[F_vsflx][o] = 0.; }
}

// inside blomgpu_step with advmth = 'remap' on the isopycnic coordinate: remap is the first stage that touches the flux arrays of
// level m and STORES (0 + flux) at every u-face of rows 0..jj+1 and every v-face of columns 0..ii+1 (stage_remap_tile.hip, zero_old),
// so only the faces beyond -- the u-points of row jj+2, the v-points of column ii+2 -- need their zeros from here
__global__ void k_init_fluxes_ring(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  const int nu = V.ii + 3, nv = V.jj + 3;
  const size_t off = (size_t)(k + mm) * V.nplane;
  if (t < nu) {
    const size_t c = IDX(V, t, V.jj + 2);
    if (V.m[I_iu][c]) { V.f[F_uflx][c + off] = 0.; V.f[F_utflx][c + off] = 0.; V.f[F_usflx][c + off] = 0.; }
  } else if (t < nu + nv) {
    const size_t c = IDX(V, V.ii + 2, t - nu);
    if (V.m[I_iv][c]) { V.f[F_vflx][c + off] = 0.; V.f[F_vtflx][c + off] = 0.; V.f[F_vsflx][c + off] = 0.; }
  }
}

int st_init_fluxes(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)nn; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  TimeScope ts(c, "init_fluxes");
  if (c->in_sequence && c->lean_fluxes && h.P.advmth == 0 && h.P.vcoord_tag == 1 && !c->remap_fold) {
    hipLaunchKernelGGL(k_init_fluxes_ring, dim3((h.ii + h.jj + 6 + 63) / 64, h.kk), dim3(64), 0, c->stream, c->d, mm);
    HIPCHK(c, hipGetLastError());
    c->fluxes_zeroed = true;
    c->fluxes_lean = true;         // the contract: the next toucher of uflx .. vsflx (m) is remap's STORING tile kernel (checked there and in blomgpu_step)
    return 0;
  }
  hipLaunchKernelGGL(k_init_fluxes, plane_grid(c->h, c->h.kk), dim3(256), 0, c->stream, c->d, mm);
  HIPCHK(c, hipGetLastError());
  c->fluxes_zeroed = c->in_sequence;
  c->fluxes_lean = false;
  return 0;
}

// ---- initms / tmsmt1, phy/mod_tmsmt.F90:161-277 ---------------------------------------------
__global__ void k_tmsmt1(const DevView *__restrict__ Vp, int off, int is_initms) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const int k = by_;                         // 0-based layer
  const size_t ok = c + (size_t)k * V.nplane, okn = c + (size_t)(k + off) * V.nplane;
  if (V.m[I_ip][c]) {
    if (!is_initms) V.f[F_dpold][okn] = V.f[F_dp][okn];
    V.f[F_told][ok] = V.f[F_temp][okn];
    V.f[F_sold][ok] = V.f[F_saln][okn];
    for (int nt = 0; nt < V.ntr; nt++)
      V.f[F_trcold][ok + (size_t)nt * V.kk * V.nplane] = V.f[F_trc][okn + (size_t)nt * 2 * V.kk * V.nplane];
  }
  if (!is_initms && V.P.vcoord_tag == 1) {
    if (V.m[I_iu][c]) V.f[F_dpuold][ok] = V.f[F_dpu][okn];
    if (V.m[I_iv][c]) V.f[F_dpvold][ok] = V.f[F_dpv][okn];
  }
}

int st_tmsmt1(blomgpu_ctx *c, int nn) {
  TimeScope ts(c, "tmsmt");
  hipLaunchKernelGGL(k_tmsmt1, plane_grid(c->h, c->h.kk), dim3(256), 0, c->stream, c->d, nn, 0);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int st_initms(blomgpu_ctx *c, int mm) {
  hipLaunchKernelGGL(k_tmsmt1, plane_grid(c->h, c->h.kk), dim3(256), 0, c->stream, c->d, mm, 1);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- p(k+1) = p(k) + dp(k+off), j,i = -2..+2 (mod_tmsmt.F90:354-365, mod_pgforc.F90:451-461,
//      mod_mxlayr.F90:1270-1280).  One thread per column, coalesced plane-by-plane. -----------
__global__ void k_pscan(const DevView *__restrict__ Vp, int off, int lo, int hi_off) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < lo || j > V.jj + hi_off || i < lo || i > V.ii + hi_off || !V.m[I_ip][c]) return;
  column_scan(V.f[F_p][c], V.f[F_dp] + (size_t)off * V.nplane + c, V.f[F_p] + c, V.nplane, V.kk);
}

// ---- the same scan with k ON THE LANES (option scan_reassoc, OFF by default: a measured tolerance-mode experiment, round 6) -------
// BASELINE.json's north star names "k-layers mapped to the wavefront ... wavefront shuffles" and "a stated floating-point
// tolerance"; the library holds tolerance zero, which forbids this kernel: a log-step prefix sum adds the layer thicknesses in a
// different order than the reference's p(k+1) = p(k) + dp(k).  A workgroup of 256 threads takes 64 consecutive points of the
// plane: all kk planes of dp are loaded coalesced (every level's load in flight at once, where the serial scan has COLUMN_U) into an
// LDS tile [k][point], each wavefront then takes every fourth column with lane = level, runs the inclusive scan over the lanes with
// six __shfl_up steps, and the tile goes back coalesced as p(2..kk+1).  kk <= 64.
#define PSL_PTS 64
#define PSL_LD (PSL_PTS + 1)        // row stride of the tile in doubles (odd: the lane-per-level reads spread over the banks)
__global__ __launch_bounds__(256) void k_pscan_lanes(const DevView *__restrict__ Vp, int off, int lo, int hi_off) {
  const DevView &V = *Vp;
  __shared__ double tile[64 * PSL_LD];
  __shared__ double base[PSL_PTS];
  __shared__ int wet[PSL_PTS];
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  (void)by_;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kk = V.kk;
  const int t0 = bx_ * PSL_PTS, t = t0 + lane;
  const size_t np = V.nplane;
  const bool inpl = t < V.nplane;
  const int i = inpl ? t % V.ni - (NBDY - 1) : 0, j = inpl ? t / V.ni - (NBDY - 1) : 0;
  const bool mine = inpl && !(j < lo || j > V.jj + hi_off || i < lo || i > V.ii + hi_off) && V.m[I_ip][inpl ? t : 0] != 0;
  gcd_t dp = V.f[F_dp] + (size_t)off * np;
  gd_t p = V.f[F_p];
  if (wv == 0) { wet[lane] = mine ? 1 : 0; base[lane] = mine ? p[t] : 0.; }
  for (int k = wv; k < kk; k += 4) tile[k * PSL_LD + lane] = mine ? dp[t + (size_t)k * np] : 0.;
  __syncthreads();
  for (int col = wv; col < PSL_PTS; col += 4) {
    if (!wet[col]) continue;                       // (wave-uniform)
    double v = lane < kk ? tile[lane * PSL_LD + col] : 0.;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double u = __shfl_up(v, d);
      if (lane >= d) v = v + u;
    }
    if (lane < kk) tile[lane * PSL_LD + col] = base[col] + v;
  }
  __syncthreads();
  if (mine)
    for (int k = wv; k < kk; k += 4) p[t + (size_t)(k + 1) * np] = tile[k * PSL_LD + lane];
}

// ---- dpu,dpv (and optionally pu,pv) from p, j,i = -1..+2 (mod_tmsmt.F90:369-391,
//      mod_pgforc.F90:463-485, mod_mxlayr.F90:1282-1310) --------------------------------------
// flags: 1 = pu, pv as well; 2 = the next step's tmsmt1 here (dpuold, dpvold = the new dpu, dpv, at the points of the tile);
// 4 = dpuold, dpvold at every point of the range (ale_regrid_remap, phy/mod_ale_regrid_remap.F90:1735-1762)
__global__ __launch_bounds__(64) void k_dpudpv(const DevView *__restrict__ Vp, int off, int flags) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < -1 || j > V.jj + 2 || i < -1 || i > V.ii + 2) return;
  const int with_pupv = flags & 1;
  const bool old_too = ((flags & 2) && j >= 1 && j <= V.jj && i >= 1 && i <= V.ii) || (flags & 4);
  const bool wu = V.m[I_iu][c] != 0, wv = V.m[I_iv][c] != 0;
  if (!wu && !wv) return;
  gcd_t p = V.f[F_p];
  const size_t np = V.nplane, w = c - 1, s = c - V.ni;
  const size_t bot = (size_t)V.kk * np;
  const double pc_b = p[c + bot];
  const double qu = wu ? fmin2(pc_b, p[w + bot]) : 0., qv = wv ? fmin2(pc_b, p[s + bot]) : 0.;
  gd_t dpu = V.f[F_dpu] + (size_t)off * np, dpv = V.f[F_dpv] + (size_t)off * np;
  double pu = wu ? V.f[F_pu][c] : 0., pv = wv ? V.f[F_pv][c] : 0.;
  double pc0 = p[c], pw0 = wu ? p[w] : 0., ps0 = wv ? p[s] : 0.;
  gd_t pug = V.f[F_pu], pvg = V.f[F_pv];
  const size_t wl = wu ? w : c, sl = wv ? s : c;           // a land neighbour's column is not read: take the own one
  const int kk = V.kk;
  for (int k0 = 0; k0 < kk; k0 += COLUMN_U) {              // COLUMN_U levels' loads in flight (blomgpu_internal.h)
    double a[COLUMN_U], b[COLUMN_U], d[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const size_t o1 = (size_t)((k0 + u < kk ? k0 + u : kk - 1) + 1) * np;
      a[u] = p[c + o1]; b[u] = p[wl + o1]; d[u] = p[sl + o1];
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 + u;
      if (k < kk) {
        const size_t o1 = (size_t)(k + 1) * np;
        const double pc1 = a[u];
        if (wu) {
          const double pw1 = b[u];
          const double dd = .5 * ((fmin2(qu, pw1) - fmin2(qu, pw0)) + (fmin2(qu, pc1) - fmin2(qu, pc0)));
          dpu[c + (size_t)k * np] = dd;
          if (old_too) V.f[F_dpuold][c + (size_t)k * np] = dd;
          if (with_pupv) { pu = pu + dd; pug[c + o1] = pu; }
          pw0 = pw1;
        }
        if (wv) {
          const double ps1 = d[u];
          const double dd = .5 * ((fmin2(qv, ps1) - fmin2(qv, ps0)) + (fmin2(qv, pc1) - fmin2(qv, pc0)));
          dpv[c + (size_t)k * np] = dd;
          if (old_too) V.f[F_dpvold][c + (size_t)k * np] = dd;
          if (with_pupv) { pv = pv + dd; pvg[c + o1] = pv; }
          ps0 = ps1;
        }
        pc0 = pc1;
      }
    }
  }
}

static void pscan_any(blomgpu_ctx *c, int off, int lo, int hi_off) {
  if (c->scan_reassoc && c->h.kk <= 64) hipLaunchKernelGGL(k_pscan_lanes, plane_grid(c->h, 1, PSL_PTS), dim3(256), 0, c->stream, c->d, off, lo, hi_off);
  else hipLaunchKernelGGL(k_pscan, plane_grid(c->h, 1, 64), dim3(64), 0, c->stream, c->d, off, lo, hi_off);
}

int launch_p_dpu_dpv(blomgpu_ctx *c, int off, int flags) {
  pscan_any(c, off, -2, 2);
  hipLaunchKernelGGL(k_dpudpv, plane_grid(c->h, 1, 64), dim3(64), 0, c->stream, c->d, off, flags);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int launch_dpudpv(blomgpu_ctx *c, int off, int flags) {
  hipLaunchKernelGGL(k_dpudpv, plane_grid(c->h, 1, 64), dim3(64), 0, c->stream, c->d, off, flags);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int launch_pscan(blomgpu_ctx *c, int off, int lo, int hi_off) {
  pscan_any(c, off, lo, hi_off);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- tmsmt2, phy/mod_tmsmt.F90:295-350: column sums then per-layer filter --------------------
// Two kernels: the rescaling factors of a column (two sums over its layers, COLUMN_U levels' loads in flight) to two
// 2-D work planes, then the filter with one thread per point and layer.  As one column kernel the filter walked the
// layers serially, one memory latency per layer: 0.28 ms on the channel grid and 0.23 ms on the three times smaller
// tripolar grid alike.
#define S2_PBFACO 0
#define S2_PBFACN 1
#define S2_PBFACM 9
// dp_in_wk (inside blomgpu_step): pbcor2 left its new, not yet rescaled layer thicknesses of level m in the work space (slot 0,
// stage_pbcor_tile.hip) and did not launch its column pass: the factor pb / p(kk+1) of that pass (phy/mod_pbcor.F90:696-726) is
// formed here with the two others, and k_tmsmt2 applies it where it reads dp of level m
__global__ __launch_bounds__(64) void k_tmsmt2_fac(const DevView *__restrict__ Vp, int m, int nn, int dp_in_wk) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  gcd_t dpo = V.f[F_dpold] + c + (size_t)nn * np, dpn = V.f[F_dp] + c + (size_t)nn * np;
  gcd_t ndp = dp_in_wk ? WK(V, 0) + c : dpn;
  double pbfaco = 0., pbfacn = 0., psum = V.f[F_p][c];
  const int kk = V.kk;
  for (int k0 = 0; k0 < kk; k0 += COLUMN_U) {
    double a0[COLUMN_U], a1[COLUMN_U], a2[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const size_t o = (size_t)(k0 + u < kk ? k0 + u : kk - 1) * np;
      a0[u] = dpo[o]; a1[u] = dpn[o]; a2[u] = dp_in_wk ? ndp[o] : 0.;
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++)
      if (k0 + u < kk) { pbfaco = pbfaco + a0[u]; pbfacn = pbfacn + a1[u]; psum = psum + a2[u]; }
  }
  const double pbm = V.f[F_pb][c + (size_t)(m - 1) * np];
  WK2(V, S2_PBFACO)[c] = pbm / pbfaco;
  WK2(V, S2_PBFACN)[c] = pbm / pbfacn;
  if (dp_in_wk) WK2(V, S2_PBFACM)[c] = pbm / psum;
}

// from_wk: pbcor2 left S, T and the tracers of level m in the work space (slots 1, 2, 3 + nt; stage_pbcor_tile.hip) instead of
// moving them into their fields: they are read there and written, filtered, to the fields
// ahead: another step follows within this call (blomgpu_step): its tmsmt1 (:230-277) would copy exactly the values written
// here -- level m of this step is level n of the next -- so they go to dpold, told, sold, trcold as well and that tmsmt1 is
// not launched (k_dpudpv does the same for dpuold, dpvold)
__global__ void k_tmsmt2(const DevView *__restrict__ Vp, int mm, int nn, int from_wk, int ahead, int dp_in_wk) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int k = by_;
  const double epsilp = 1.e-12;
  const double pbfaco = WK2(V, S2_PBFACO)[c], pbfacn = WK2(V, S2_PBFACN)[c];
  const double wts1 = V.P.wts1, wts2 = V.P.wts2;
  const size_t okm = c + (size_t)(k + mm) * np, okn = c + (size_t)(k + nn) * np, ok = c + (size_t)k * np;
  double pold = fmax2(0., V.f[F_dpold][okn] * pbfaco);
  double pmid = fmax2(0., dp_in_wk ? WK(V, 0)[ok] * WK2(V, S2_PBFACM)[c] : V.f[F_dp][okm]);
  double pnew = fmax2(0., V.f[F_dp][okn] * pbfacn);
  const double dpm = wts1 * pmid + wts2 * (pold + pnew);
  V.f[F_dp][okm] = dpm;
  pold = pold + epsilp;
  pmid = pmid + epsilp;
  pnew = pnew + epsilp;
  const double tmid = from_wk ? WK(V, 2)[ok] : V.f[F_temp][okm], smid = from_wk ? WK(V, 1)[ok] : V.f[F_saln][okm];
  const double tnew = (wts1 * pmid * tmid + wts2 * (pold * V.f[F_told][ok] + pnew * V.f[F_temp][okn])) / (dpm + epsilp);
  const double snew = (wts1 * pmid * smid + wts2 * (pold * V.f[F_sold][ok] + pnew * V.f[F_saln][okn])) / (dpm + epsilp);
  V.f[F_temp][okm] = tnew;
  V.f[F_saln][okm] = snew;
  if (ahead) { V.f[F_dpold][okm] = dpm; V.f[F_told][ok] = tnew; V.f[F_sold][ok] = snew; }
  for (int nt = 0; nt < V.ntr; nt++) {
    gd_t tr = V.f[F_trc] + (size_t)nt * 2 * V.kk * np;
    gd_t tro = V.f[F_trcold] + (size_t)nt * V.kk * np;
    const double xmid = from_wk ? WK(V, 3 + nt)[ok] : tr[okm];
    const double xnew = (wts1 * pmid * xmid + wts2 * (pold * tro[ok] + pnew * tr[okn])) / (dpm + epsilp);
    tr[okm] = xnew;
    if (ahead) tro[ok] = xnew;
  }
}

// ahead, arctic patch: the halo update that follows the filter rewrites the seam row jj of dp(km), and the next step's tmsmt1
// would copy what it left there
__global__ void k_tmsmt_dpold_seam(const DevView *__restrict__ Vp, int mm) {
  const DevView &V = *Vp;
  const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, k = blockIdx.y;
  if (i > V.ii) return;
  const size_t c = IDX(V, i, V.jj);
  if (!V.m[I_ip][c]) return;
  const size_t okm = c + (size_t)(k + mm) * V.nplane;
  V.f[F_dpold][okm] = V.f[F_dp][okm];
}

int st_tmsmt2(blomgpu_ctx *c, int m, int mm, int nn, int k1m) {
  TimeScope ts(c, "tmsmt");
  const int dp_in_wk = c->pbcor2_dp_in_wk ? 1 : 0;
  c->pbcor2_dp_in_wk = false;
  hipLaunchKernelGGL(k_tmsmt2_fac, plane_grid(c->h, 1, 64), dim3(64), 0, c->stream, c->d, m, nn, dp_in_wk);
  const int from_wk = c->pbcor2_handed_over ? 1 : 0;
  c->pbcor2_handed_over = false;
  const int ahead = c->in_sequence && c->tmsmt1_ahead ? 1 : 0;
  hipLaunchKernelGGL(k_tmsmt2, plane_grid(c->h, c->h.kk), dim3(256), 0, c->stream, c->d, mm, nn, from_wk, ahead, dp_in_wk);
  HIPCHK(c, hipGetLastError());
  c->tmsmt1_done_ahead = ahead != 0;
  if (int rc = st_xctilr(c, c->h.f[F_dp] + (size_t)(k1m - 1) * c->h.nplane, 1, c->h.kk, 3, 3, 1)) return rc;
  if (ahead && c->h.nreg == 2)
    hipLaunchKernelGGL(k_tmsmt_dpold_seam, dim3((c->h.ii + 63) / 64, c->h.kk), dim3(64), 0, c->stream, c->d, mm);
  if (c->h.P.vcoord_tag == 1) return launch_p_dpu_dpv(c, mm, ahead ? 2 : 0);
  return launch_pscan(c, mm, -2, 2);
}

// ---- tail of mxlayr, phy/mod_mxlayr.F90:1266-1310 -------------------------------------------
int st_mxlayr_tail(blomgpu_ctx *c, int nn, int k1n) {
  if (int rc = st_xctilr(c, c->h.f[F_dp] + (size_t)(k1n - 1) * c->h.nplane, 1, c->h.kk, 3, 3, 1)) return rc;
  return launch_p_dpu_dpv(c, nn, 0);
}

// ---- kfpla halo through util1, phy/mod_cmnfld_routines.F90:1176-1196 --------------------------
__global__ void k_kfpla_util(const DevView *__restrict__ Vp, int n, int back) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (!V.m[I_ip][c]) return;
  gi_t kf = V.m[I_kfpla] + (size_t)(n - 1) * V.nplane;
  if (!back) {
    if (j >= 1 && j <= V.jj && i >= 1 && i <= V.ii) V.f[F_util1][c] = (double)kf[c];
  } else if (j >= -1 && j <= V.jj + 2 && i >= -1 && i <= V.ii + 2)
    kf[c] = (int)lround(V.f[F_util1][c]);
}

int st_kfpla_halo(blomgpu_ctx *c, int n) {
  hipLaunchKernelGGL(k_kfpla_util, plane_grid(c->h), dim3(256), 0, c->stream, c->d, n, 0);
  if (int rc = st_xctilr(c, c->h.f[F_util1], 1, 1, 2, 2, 1)) return rc;
  hipLaunchKernelGGL(k_kfpla_util, plane_grid(c->h), dim3(256), 0, c->stream, c->d, n, 1);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- updtrc, trc/mod_tracers_update.F90:152-170 = hamocc_step (iHAMOCC, not on this path) + idlage_step,
//      idlage/mod_idlage.F90:57-96: surface layer reset, deeper layers aged, all p-points incl. the halo ----
__global__ void k_idlage_step(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (!V.m[I_ip][c]) return;
  const int k = by_;
  const size_t np = V.nplane;
  gd_t t = V.f[F_trc] + c + ((size_t)(k + nn) + (size_t)(V.P.itriag - 1) * 2 * V.kk) * np;
  if (k == 0) *t = 0.;
  else *t = *t + V.P.delt1 / (86400. * V.P.nday_in_year);
}

int st_updtrc(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.P.itriag < 1 || h.P.itriag > h.ntr) return 0;       // no ideal age tracer configured
  if (h.P.nday_in_year < 1) return ctx_fail(c, "updtrc: nday_in_year is not set (mod_time)");
  // inside blomgpu_step (phys_dag) on the second stream, beside barotp, which reads and writes no tracer; blomgpu_step waits in front of pbcor2
  const bool aside = ctx_overlap_on(c) && (c->phys_dag & 4) && !c->tiling.multi();
  if (aside)
    if (int rc = ctx_side_fork(c, 8)) return rc;
  TimeScope ts(c, "updtrc");
  hipLaunchKernelGGL(k_idlage_step, plane_grid(h, h.kk), dim3(256), 0, aside ? c->side : c->stream, c->d, nn);
  HIPCHK(c, hipGetLastError());
  if (aside) {
    if (int rc = ctx_side_done(c, 9)) return rc;
    c->updtrc_on_side = true;
  }
  return 0;
}

// ---- budget_sums, phy/mod_budget.F90:95-196 (use_TRC; no GLS): mass weighted column sums of S and T (and the TKE
//      tracer, util3) into util1, util2 (which = 0) or of tracer 1 into util1 (which = 1); their global sums through xcsum.  The salt
//      correction term of call 5 (:182-194) needs mod_forcing's salt_corr, which no stage here produces. ----
__global__ void k_budget_columns(const DevView *__restrict__ Vp, int nn, int which) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const double scp2 = V.f[F_scp2][c];
  double s1 = 0., s2 = 0., s3 = 0.;
  const bool tke = which == 0 && V.P.itrtke >= 1;
  for (int k = 0; k < V.kk; k++) {
    const size_t o = c + (size_t)(k + nn) * np;
    const double q = V.f[F_dp][o] * scp2;
    if (which == 0) {
      s1 = s1 + V.f[F_saln][o] * q;
      s2 = s2 + V.f[F_temp][o] * q;
      if (tke) s3 = s3 + V.f[F_trc][o + (size_t)(V.P.itrtke - 1) * 2 * V.kk * np] * q;
    } else {
      s1 = s1 + V.f[F_trc][o] * q;
    }
  }
  V.f[F_util1][c] = s1;
  if (which == 0) V.f[F_util2][c] = s2;
  if (tke) V.f[F_util3][c] = s3;
}

int st_budget_sums(blomgpu_ctx *c, int ncall, int n, int nn) {
  if (!c->cnsvdi) return 0;                                     // :105
  if (ncall < 1 || ncall > 7 || n < 1 || n > 2) return ctx_fail(c, "budget_sums: ncall in 1..7, n in 1..2");
  const DevView &h = c->h;
  hipLaunchKernelGGL(k_budget_columns, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn, 0);
  if (int rc = st_xcsum(c, h.f[F_util1], 1, &c->budget[0][ncall - 1][n - 1])) return rc;
  if (int rc = st_xcsum(c, h.f[F_util2], 1, &c->budget[1][ncall - 1][n - 1])) return rc;
  if (h.P.itrtke >= 1)
    if (int rc = st_xcsum(c, h.f[F_util3], 1, &c->budget[3][ncall - 1][n - 1])) return rc;
  if (h.ntr >= 1) {
    hipLaunchKernelGGL(k_budget_columns, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn, 1);
    if (int rc = st_xcsum(c, h.f[F_util1], 1, &c->budget[2][ncall - 1][n - 1])) return rc;
  }
  return 0;
}
