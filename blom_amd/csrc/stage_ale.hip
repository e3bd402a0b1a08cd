// ale_regrid_remap -- phy/mod_ale_regrid_remap.F90:1486-1984, first piece of SURVEY.md 8 row f3 (the vertical coordinates
// other than isopyc_bulkml): regrid the layer interfaces, remap T, S, the tracers and the velocities to the new layers.
//
// The reference walks the tile row by row ("j-slices") and calls mod_hor3map once per column and tracer
// (reconstruct_trc_jslice :186-261, regrid_*_jslice :263-944, remap_trc_jslice :1022-1057, copy_jslice_to_3d :1153-1179, then
// the same for the u- and the v-columns :1696-1900).  Here the engine is the device hor3map of this library
// (hor3map.hip, include/blomgpu_hor3map.h -- pinned on the reference's module) called once per SLAB: its internal layout,
// [level][column] with the column fastest, IS the layout of the model's fields (one level = one padded plane), so the
// fields go in and out without a transpose (io mode 2).  A column of the slab is a point of the padded plane; points that
// are not wet points of the tile get a harmless column (unit layers) whose result is never copied back.
//
// Built: vcoord_type = 'plevel' (regrid_plevel_jslice, :263-284): interfaces at prescribed pressures below the surface; and
// vcoord_type = 'cntiso_hybrid' with regrid_method = 'direct' (regrid_cntiso_hybrid_direct_jslice, :286-558): interfaces where
// the monotonically reconstructed potential density takes the layers' target values, bounded by minimum thicknesses and blended
// into the pressure levels towards the surface.
// Not built yet: regrid_method = 'nudge' (:560-916) with its lateral smoothing (:946-1020), neutral diffusion (mod_ndiff), the
// z-level diagnostics (remap_trc_diazlv_jslice) -- each fails loudly.
// Parity: cross-checked against the reference's REAL module compiled against a stand-in for mod_dia (oracle/Makefile
// *_xale, tests/test_xcheck_ale.py) -- a cross-check, not a pin.
#include "blomgpu_internal.h"
#include "eos.h"
#include "../../include/blomgpu_hor3map.h"

int h3m_use_stream(blomgpu_h3m_grid *G, hipStream_t stream);          // hor3map.hip

#define H3M_MAXF 8

struct AleState {
  blomgpu_h3m_grid *grid = nullptr;
  std::vector<blomgpu_h3m_src *> trc;       // T, S, tracers (init_ale_regrid_remap :1412-1432)
  blomgpu_h3m_src *vel = nullptr, *dens = nullptr;
  blomgpu_h3m_map *map = nullptr;
  double *plane = nullptr;                  // scratch: p_src, p_dst (kk+1 planes each), remapped fields (kk planes each)
  size_t plane_n = 0;
  int ntr_loc = 0, method = 0;
};

void ale_free(blomgpu_ctx *c) {
  AleState *a = (AleState *)c->ale;
  if (!a) return;
  if (a->grid) blomgpu_h3m_grid_free(a->grid);             // frees the sources and the map too
  if (a->plane) (void)hipFree(a->plane);
  delete a;
  c->ale = nullptr;
}

static int ale_fail(blomgpu_ctx *c, const char *what, int rc) {
  return ctx_fail(c, (std::string("ale_regrid_remap: ") + what + ": " + (rc > 0 ? blomgpu_h3m_errstr(rc) : "device layer failure")).c_str());
}

// the structures of init_ale_regrid_remap (:1357-1484); options as read by readnml_ale_regrid_remap (:1185-1355)
static int ale_prepare(blomgpu_ctx *c) {
  const DevView &h = c->h;
  AleState *a = (AleState *)c->ale;
  const int ntr_loc = h.ntr + 2;
  if (a && (a->ntr_loc != ntr_loc || a->method != c->ale_method)) { ale_free(c); a = nullptr; }
  if (a) return 0;
  a = new AleState;
  c->ale = a;
  a->ntr_loc = ntr_loc; a->method = c->ale_method;
  int rc = blomgpu_h3m_grid_create(&a->grid, c->device, (int)h.nplane, h.kk, c->ale_method, c->ale_upper_bndr_ord, c->ale_lower_bndr_ord);
  if (rc) return ale_fail(c, "initialize_rcgs", rc);
  if ((rc = h3m_use_stream(a->grid, c->stream))) return ale_fail(c, "stream", rc);
  (void)blomgpu_h3m_set_io(a->grid, 2, 1);
  // tracer_limiting = non_oscillatory: T keeps it, S and the tracers get non_oscillatory_posdef (:1419-1431)
  for (int nt = 0; nt < ntr_loc; nt++) {
    int lim = c->ale_tracer_limiting;
    if (nt > 0 && lim == BLOMGPU_H3M_NON_OSCILLATORY) lim = BLOMGPU_H3M_NON_OSCILLATORY_POSDEF;
    blomgpu_h3m_src *s = nullptr;
    if ((rc = blomgpu_h3m_src_create(a->grid, &s, lim, c->ale_tracer_pc_upper, c->ale_tracer_pc_lower))) return ale_fail(c, "initialize_rcss", rc);
    a->trc.push_back(s);
  }
  if ((rc = blomgpu_h3m_src_create(a->grid, &a->vel, c->ale_velocity_limiting, c->ale_velocity_pc_upper, c->ale_velocity_pc_lower)))
    return ale_fail(c, "initialize_rcss", rc);
  // potential density: density_limiting = 'monotonic' is the only value the reference accepts (:1287-1297)
  if ((rc = blomgpu_h3m_src_create(a->grid, &a->dens, BLOMGPU_H3M_MONOTONIC, c->ale_density_pc_upper, c->ale_density_pc_lower)))
    return ale_fail(c, "initialize_rcss", rc);
  if ((rc = blomgpu_h3m_map_create(a->grid, &a->map, h.kk))) return ale_fail(c, "initialize_rms", rc);
  // p_src, p_dst | remapped fields | polynomial coefficients of T and S (5 per layer at most) | sig_src | sig_trg
  a->plane_n = ((size_t)2 * (h.kk + 1) + (size_t)H3M_MAXF * h.kk + (size_t)10 * h.kk + h.kk + (h.kk + 1)) * h.nplane;
  HIPCHK(c, hipMalloc((void **)&a->plane, sizeof(double) * a->plane_n));
  return 0;
}

#define PLANE_T(V)                                                         \
  const int t_ = blockIdx.x * blockDim.x + threadIdx.x;                    \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_

// source interfaces of the p-columns (:203-211) and their regridded positions for vcoord_type = 'plevel' (:263-284);
// any other point of the plane: unit layers, left where they are
__global__ void k_ale_p_src_dst(const DevView *__restrict__ Vp, int nn, const double *__restrict__ plevel, double *__restrict__ psrc,
                                double *__restrict__ pdst) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 && j <= V.jj && i >= 1 && i <= V.ii && V.m[I_ip][c];
  if (!col) {
    for (int k = 0; k <= kk; k++) { psrc[c + (size_t)k * np] = (double)k; pdst[c + (size_t)k * np] = (double)k; }
    return;
  }
  const double p1 = V.f[F_p][c];
  double acc = p1;
  psrc[c] = acc;
  for (int k = 0; k < kk; k++) {
    acc = acc + V.f[F_dp][c + (size_t)(k + nn) * np];
    psrc[c + (size_t)(k + 1) * np] = acc;
  }
  const double pbot = acc;
  for (int k = 0; k < kk; k++) pdst[c + (size_t)k * np] = fmin2(plevel[k] + p1, pbot);
  pdst[c + (size_t)kk * np] = pbot;
}

// ---- vcoord_type = 'cntiso_hybrid', regrid_method = 'direct': regrid_cntiso_hybrid_direct_jslice, :286-558 --------------------
// 1-based accessors of a column's planes
#define PL(a, k) (a)[c + (size_t)((k)-1) * np]
#define ALE_BFSQ_MIN 1.e-7
#define ALE_GRAV 9.806
#define ALE_EPSILP 1.e-12
#define ALE_MVAL (-1.e33)

// first half, :311-404: interface densities of the reconstructed T, S (the diagnostic array sigint), then the layer densities
// made monotonic in depth for the reconstruction that regrid() inverts.  pcT, pcS: the polynomial coefficients of the T and S
// reconstructions (npc per layer: extract_polycoeff); peval1 (:152-160) sums FIVE coefficients, the absent ones are zero.
__global__ void k_ale_direct_pre(const DevView *__restrict__ Vp, int nn, const double *__restrict__ psrc, const double *__restrict__ pcT,
                                 const double *__restrict__ pcS, int npc, double *__restrict__ sgs, double *__restrict__ sgt) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 && j <= V.jj && i >= 1 && i <= V.ii && V.m[I_ip][c];
  if (!col) {                               // harmless column: densities that increase with depth, targets between them
    for (int k = 1; k <= kk; k++) { PL(sgs, k) = (double)k; PL(sgt, k) = (double)k - .5; }
    PL(sgt, kk + 1) = (double)kk + .5;
    return;
  }
  int ksmx = kk;                                                                     // :213-217
  for (int k = kk; k >= 1; k--)
    if (PL(psrc, k) == PL(psrc, kk + 1)) ksmx = k - 1;
  auto top = [&](const double *pc, int k) { return pc[c + (size_t)(npc * (k - 1)) * np]; };            // peval0
  auto bot = [&](const double *pc, int k) {                                                            // peval1
    double f = pc[c + (size_t)(npc * (k - 1)) * np];
    for (int q = 1; q < 5; q++) f = f + (q < npc ? pc[c + (size_t)(npc * (k - 1) + q) * np] : 0.);
    return f;
  };
  double *sigint = V.f[F_sigint];
  double sd2_prev = 0.;
  for (int k = 1; k <= ksmx; k++) {                                                   // :317-327
    const double sd1 = eos::sig(V.P, top(pcT, k), top(pcS, k));
    const double sd2 = eos::sig(V.P, bot(pcT, k), bot(pcS, k));
    PL(sigint, k) = k == 1 ? sd1 : .5 * (sd2_prev + sd1);
    sd2_prev = sd2;
  }
  for (int k = ksmx + 1; k <= kk; k++) PL(sigint, k) = sd2_prev;
  const double *sigma = V.f[F_sigma] + (size_t)nn * np, *sigmar = V.f[F_sigmar];
  for (int k = 1; k <= kk; k++) { PL(sgs, k) = PL(sigma, k); PL(sgt, k) = PL(sigmar, k); }              // :330-335
  PL(sgt, kk + 1) = PL(sgt, kk);
  const double beta = ALE_BFSQ_MIN / (ALE_GRAV * ALE_GRAV);
#define P(k) PL(psrc, k)
#define S(k) PL(sgs, k)
  int kl = kk, ku = kl - 1;                                                           // :339-398
  while (ku > 0) {
    bool thin = P(kl + 1) - P(ku) < ALE_EPSILP;
    if (thin || S(kl) - S(ku) < .5 * beta * (P(kl + 1) - P(ku))) {
      double sdpsum = S(ku) * (P(ku + 1) - P(ku)) + S(kl) * (P(kl + 1) - P(kl));
      double smean = 0.;
      if (!thin) smean = sdpsum / (P(kl + 1) - P(ku));
      while (true) {
        bool added = false;
        if (ku > 1) {
          if (thin) {
            ku = ku - 1;
            sdpsum = sdpsum + S(ku) * (P(ku + 1) - P(ku));
            thin = P(kl + 1) - P(ku) < ALE_EPSILP;
            if (!thin) smean = sdpsum / (P(kl + 1) - P(ku));
            added = true;
          } else if (smean - S(ku - 1) < .5 * beta * (P(kl + 1) - P(ku - 1))) {
            ku = ku - 1;
            sdpsum = sdpsum + S(ku) * (P(ku + 1) - P(ku));
            smean = sdpsum / (P(kl + 1) - P(ku));
            added = true;
          }
        }
        if (kl < kk) {
          if (thin) {
            kl = kl + 1;
            sdpsum = sdpsum + S(kl) * (P(kl + 1) - P(kl));
            thin = P(kl + 1) - P(ku) < ALE_EPSILP;
            if (!thin) smean = sdpsum / (P(kl + 1) - P(ku));
            added = true;
          } else if (S(kl + 1) - smean < .5 * beta * (P(kl + 2) - P(ku))) {
            kl = kl + 1;
            sdpsum = sdpsum + S(kl) * (P(kl + 1) - P(kl));
            smean = sdpsum / (P(kl + 1) - P(ku));
            added = true;
          }
        }
        if (!added) break;
      }
      for (int k = ku; k <= kl; k++) S(k) = smean + .5 * beta * (P(k) + P(k + 1) - P(ku) - P(kl + 1));
    }
    kl = ku;
    ku = kl - 1;
  }
#undef S
}

// second half, :419-553: bound the regridded interfaces by the water column, keep the layers above a minimum thickness,
// and blend into the prescribed pressure levels towards the surface
__global__ void k_ale_direct_post(const DevView *__restrict__ Vp, const double *__restrict__ psrc, const double *__restrict__ sgs,
                                  const double *__restrict__ sgt, const double *__restrict__ plevel, double dpmin_interior,
                                  int k_range_plevel, double *__restrict__ pdst) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 && j <= V.jj && i >= 1 && i <= V.ii && V.m[I_ip][c];
  if (!col) {
    for (int k = 1; k <= kk + 1; k++) PL(pdst, k) = PL(psrc, k);
    return;
  }
#define D(k) PL(pdst, k)
  int k = 1, ks, ke;                                                                  // :424-440
  while (true) {
    ks = k;
    if (D(k) != ALE_MVAL) break;
    D(k) = P(1);
    if (k > kk) break;
    k = k + 1;
  }
  k = kk + 1;
  while (true) {
    ke = k;
    if (D(k) != ALE_MVAL) break;
    D(k) = P(kk + 1);
    if (k == 1) break;
    k = k - 1;
  }
  D(1) = P(1);
  D(kk + 1) = P(kk + 1);
  if (ks == ke) {                                                                     // :445-460
    double sdpsum = 0.;
    for (int q = 1; q <= kk; q++) sdpsum = sdpsum + PL(sgs, q) * (P(q + 1) - P(q));
    const double smean = sdpsum / (P(kk + 1) - P(1));
    ks = 2;
    while (ks <= kk) {
      if (smean < PL(sgt, ks)) break;
      ks = ks + 1;
    }
    for (int q = ks; q <= kk; q++) D(q) = P(kk + 1);
    ke = ks - 1;
  }
  const double dpmin = fmin2(plevel[1] - plevel[0], dpmin_interior);                 // :465
  ks = ks > 2 ? ks : 2;
  ke = ke < kk ? ke : kk;
  k = ks;
  while (k <= ke) {                                                                   // :468-524
    if (D(k + 1) - D(k) < dpmin) {
      if (k == ke) D(k) = D(ke + 1);
      else {
        int ku = k, kl = k + 1;
        double pku = .5 * (D(kl) + D(ku) - dpmin);
        while (true) {
          bool added = false;
          kl = kl + 1;
          double pku_test = ((pku - dpmin) * (kl - ku) + D(kl)) / (kl - ku + 1);
          if (pku_test + (kl - ku) * dpmin > D(kl)) {
            if (kl == ke + 1) break;
            pku = pku_test;
            added = true;
          } else kl = kl - 1;
          ku = ku - 1;
          pku_test = ((pku - dpmin) * (kl - ku) + D(ku)) / (kl - ku + 1);
          if (pku_test < D(ku)) {
            if (ku == 1) break;
            pku = pku_test;
            added = true;
          } else ku = ku + 1;
          if (!added) break;
        }
        if (ku == 1) {
          for (int q = 2; q <= kl; q++) D(q) = fmin2(D(ke + 1), D(q - 1) + dpmin);
          for (int q = kl + 1; q <= ke; q++) D(q) = fmin2(D(ke + 1), fmax2(D(q), D(1) + dpmin * (q - 1)));
        } else if (kl == ke + 1) {
          for (int q = ku; q <= kl; q++) D(q) = D(ke + 1);
        } else {
          D(ku) = pku;
          for (int q = ku + 1; q <= kl; q++) D(q) = D(q - 1) + dpmin;
        }
        k = kl;
      }
    }
    k = k + 1;
  }
#define LEV(q) plevel[(q)-1]
  for (int q = 2; q <= k_range_plevel; q++) D(q) = fmin2(D(kk + 1), LEV(q) + P(1));   // :530-532
  double dpt = LEV(k_range_plevel + 1) - LEV(k_range_plevel);
  for (int q = k_range_plevel + 1; q <= ke; q++) {                                     // :534-553
    const double pmin = LEV(q) + P(1);
    const int qq = (q < kk - 1 ? q : kk - 1);
    dpt = fmax2(fmax2(D(q + 1) - D(q), dpt), LEV(qq + 1) - LEV(qq));
    double pt = fmax2(D(q), pmin);
    const double ptu1 = pmin - dpt, ptl1 = pmin + dpt, ptu2 = pmin, ptl2 = pmin + 2. * dpt;
    const double w1 = fmin2(1., (D(q) - P(1)) / (pmin - P(1)));
    if (D(q) > ptu1 && D(q) < ptl1) {
      const double x = .5 * (D(q) - ptu1) / dpt;
      pt = pmin + dpt * x * x;
    }
    if (D(q + 1) > ptu2 && D(q + 1) < ptl2) {
      const double x = .5 * (D(q + 1) - ptu2) / dpt;
      pt = w1 * pt + (1. - w1) * (pmin + dpt * x * x);
    }
    D(q) = fmin2(D(ke + 1), fmax2(D(q - 1) + dpmin, pt));
  }
#undef LEV
#undef D
#undef P
}

// copy_jslice_to_3d (:1153-1179) for up to H3M_MAXF remapped fields starting with field f0 of (T, S, tracer 1, ..)
__global__ void k_ale_copy_back(const DevView *__restrict__ Vp, int nn, const double *__restrict__ pdst, const double *__restrict__ rm,
                                int f0, int nf) {
  const DevView &V = *Vp;
  PLANE_T(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, per = (size_t)V.kk * np;
  const int k = blockIdx.y;
  const size_t okn = c + (size_t)(k + nn) * np, ok = c + (size_t)k * np;
  for (int f = 0; f < nf; f++) {
    const int nt = f0 + f;                                 // 0: T, 1: S, 2..: tracers
    const double v = rm[(size_t)f * per + ok];
    if (nt == 0) V.f[F_temp][okn] = v;
    else if (nt == 1) V.f[F_saln][okn] = v;
    else V.f[F_trc][okn + (size_t)(nt - 2) * 2 * V.kk * np] = v;
  }
  if (f0 == 0) {                                           // T and S travel in the first group: dp and sigma with them
    V.f[F_dp][okn] = pdst[c + (size_t)(k + 1) * np] - pdst[ok];
    V.f[F_sigma][okn] = eos::sig(V.P, rm[ok], rm[per + ok]);
  }
}

// :1696-1711 pu, pv from the OLD dpu, dpv of the interior velocity points
__global__ void k_ale_pupv(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_T(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const size_t np = V.nplane;
  if (V.m[I_iu][c]) {
    double a = V.f[F_pu][c];
    for (int k = 0; k < V.kk; k++) { a = a + V.f[F_dpu][c + (size_t)(k + nn) * np]; V.f[F_pu][c + (size_t)(k + 1) * np] = a; }
  }
  if (V.m[I_iv][c]) {
    double a = V.f[F_pv][c];
    for (int k = 0; k < V.kk; k++) { a = a + V.f[F_dpv][c + (size_t)(k + nn) * np]; V.f[F_pv][c + (size_t)(k + 1) * np] = a; }
  }
}

// :1715-1733 the old bottom pressure into util1 (j = -2..jj+3, i = -1..ii), then p from the new dp (i = -2..ii+3)
__global__ void k_ale_pscan(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_T(V);
  if (j < -2 || j > V.jj + 3 || i < -2 || i > V.ii + 3 || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  if (i >= -1 && i <= V.ii) V.f[F_util1][c] = V.f[F_p][c + (size_t)V.kk * np];
  double a = V.f[F_p][c];
  for (int k = 0; k < V.kk; k++) { a = a + V.f[F_dp][c + (size_t)(k + nn) * np]; V.f[F_p][c + (size_t)(k + 1) * np] = a; }
}

// source and destination interfaces of the u- (isv = 0) or v-columns (:1768-1779, :1836-1847): the source interfaces are the
// old ones rescaled to the new depth of the column
__global__ void k_ale_uv_src_dst(const DevView *__restrict__ Vp, int nn, int isv, double *__restrict__ psrc, double *__restrict__ pdst) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 && j <= V.jj && i >= 1 && i <= V.ii && (isv ? V.m[I_iv][c] : V.m[I_iu][c]);
  if (!col) {
    for (int k = 0; k <= kk; k++) { psrc[c + (size_t)k * np] = (double)k; pdst[c + (size_t)k * np] = (double)k; }
    return;
  }
  const double *pz = isv ? V.f[F_pv] : V.f[F_pu], *dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  double a = pz[c];
  pdst[c] = a;
  for (int k = 0; k < kk; k++) { a = a + dpz[c + (size_t)k * np]; pdst[c + (size_t)(k + 1) * np] = a; }
  const double *u1 = V.f[F_util1];
  const double q = fmin2(u1[isv ? c - V.ni : c - 1], u1[c]) / pz[c + (size_t)kk * np];
  for (int k = 0; k <= kk; k++) psrc[c + (size_t)k * np] = pz[c + (size_t)k * np] * q;
}

__global__ void k_ale_uv_back(const DevView *__restrict__ Vp, int nn, int isv, const double *__restrict__ rm) {
  const DevView &V = *Vp;
  PLANE_T(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const int k = blockIdx.y;
  (isv ? V.f[F_v] : V.f[F_u])[c + (size_t)(k + nn) * V.nplane] = rm[c + (size_t)k * V.nplane];
}

int st_ale_regrid_remap(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)mm; (void)k1m;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "ale_regrid_remap: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:138-149)");
  if (h.P.vcoord_tag == 2 && c->ale_regrid_method != 1)
    return ctx_fail(c, "ale_regrid_remap: vcoord_type = 'cntiso_hybrid' is built with regrid_method = 'direct' only ('nudge' and its "
                       "lateral smoothing, phy/mod_ale_regrid_remap.F90:560-1020, are not)");
  if (h.P.vcoord_tag != 2 && h.P.vcoord_tag != 3) return ctx_fail(c, "ale_regrid_remap: unknown vertical coordinate");
  if (h.P.ltedtp_opt != 1) return ctx_fail(c, "ale_regrid_remap: neutral diffusion (ltedtp = 'neutral', phy/mod_ndiff.F90) is not built");
  if (!c->ale_plevel) return ctx_fail(c, "ale_regrid_remap: the pressure levels are not set (blomgpu_set_vector \"plevel\", phy/mod_vcoord.F90:99)");
  if (c->tiling.multi()) return ctx_fail(c, "ale_regrid_remap: not built for decomposed domains yet");
  if (int rc = ale_prepare(c)) return rc;
  AleState *a = (AleState *)c->ale;
  const size_t np = h.nplane, per = (size_t)h.kk * np;
  double *psrc = a->plane, *pdst = psrc + (size_t)(h.kk + 1) * np, *rm = pdst + (size_t)(h.kk + 1) * np;
  const dim3 g1((unsigned)((np + 255) / 256)), gk((unsigned)((np + 255) / 256), h.kk), b(256);
  int rc;
  // ---- tracers ----------------------------------------------------------------------------------------------------------
  hipLaunchKernelGGL(k_ale_p_src_dst, g1, b, 0, c->stream, c->d, nn, (const double *)c->ale_plevel, psrc, pdst);
  if ((rc = blomgpu_h3m_prepare_reconstruction(a->grid, psrc))) return ale_fail(c, "prepare_reconstruction", rc);
  if (h.P.vcoord_tag == 2) {
    // regrid_cntiso_hybrid_direct_jslice: the interfaces go where the reconstructed potential density takes its target values
    double *pcT = rm + (size_t)H3M_MAXF * per, *pcS = pcT + (size_t)5 * per, *sgs = pcS + (size_t)5 * per, *sgt = sgs + per;
    const double *ts[2] = {h.f[F_temp] + (size_t)nn * np, h.f[F_saln] + (size_t)nn * np};
    blomgpu_h3m_src *tss[2] = {a->trc[0], a->trc[1]};
    if ((rc = blomgpu_h3m_reconstruct_many(a->grid, 2, tss, ts))) return ale_fail(c, "reconstruct", rc);
    if ((rc = blomgpu_h3m_extract_polycoeff(a->trc[0], pcT))) return ale_fail(c, "extract_polycoeff", rc);
    if ((rc = blomgpu_h3m_extract_polycoeff(a->trc[1], pcS))) return ale_fail(c, "extract_polycoeff", rc);
    const int npc = c->ale_method == BLOMGPU_H3M_PLM ? 2 : (c->ale_method == BLOMGPU_H3M_PPM ? 3 : 5);
    hipLaunchKernelGGL(k_ale_direct_pre, g1, b, 0, c->stream, c->d, nn, (const double *)psrc, (const double *)pcT, (const double *)pcS,
                       npc, sgs, sgt);
    if ((rc = blomgpu_h3m_reconstruct(a->grid, a->dens, sgs))) return ale_fail(c, "reconstruct (density)", rc);
    if ((rc = blomgpu_h3m_regrid(a->dens, h.kk + 1, sgt, pdst, -1.e33, 0))) return ale_fail(c, "regrid", rc);
    hipLaunchKernelGGL(k_ale_direct_post, g1, b, 0, c->stream, c->d, (const double *)psrc, (const double *)sgs, (const double *)sgt,
                       (const double *)c->ale_plevel, c->ale_dpmin_interior, c->ale_k_range_plevel, pdst);
  }
  if ((rc = blomgpu_h3m_prepare_remapping(a->grid, a->map, pdst))) return ale_fail(c, "prepare_remapping", rc);
  for (int f0 = 0; f0 < a->ntr_loc; f0 += H3M_MAXF) {
    const int nf = a->ntr_loc - f0 < H3M_MAXF ? a->ntr_loc - f0 : H3M_MAXF;
    const double *us[H3M_MAXF];
    double *ud[H3M_MAXF];
    blomgpu_h3m_src *ss[H3M_MAXF];
    for (int f = 0; f < nf; f++) {
      const int nt = f0 + f;
      us[f] = nt == 0 ? h.f[F_temp] + (size_t)nn * np
            : nt == 1 ? h.f[F_saln] + (size_t)nn * np : h.f[F_trc] + ((size_t)nn + (size_t)(nt - 2) * 2 * h.kk) * np;
      ud[f] = rm + (size_t)f * per;
      ss[f] = a->trc[nt];
    }
    if ((rc = blomgpu_h3m_reconstruct_many(a->grid, nf, ss, us))) return ale_fail(c, "reconstruct", rc);
    if ((rc = blomgpu_h3m_remap_many(nf, ss, a->map, ud))) return ale_fail(c, "remap", rc);
    hipLaunchKernelGGL(k_ale_copy_back, gk, b, 0, c->stream, c->d, nn, (const double *)pdst, (const double *)rm, f0, nf);
  }
  // ---- velocities, :1692-1900 -------------------------------------------------------------------------------------------------
  hipLaunchKernelGGL(k_ale_pupv, g1, b, 0, c->stream, c->d, nn);
  if (int rc2 = st_xctilr(c, h.f[F_dp] + (size_t)(k1n - 1) * np, 1, h.kk, 3, 3, 1)) return rc2;
  hipLaunchKernelGGL(k_ale_pscan, g1, b, 0, c->stream, c->d, nn);
  // dpu, dpv of the new layers and their copies dpuold, dpvold, j,i = -1..+2 (:1735-1762)
  if (int rc2 = launch_dpudpv(c, nn, 4)) return rc2;
  for (int isv = 0; isv < 2; isv++) {
    hipLaunchKernelGGL(k_ale_uv_src_dst, g1, b, 0, c->stream, c->d, nn, isv, psrc, pdst);
    if ((rc = blomgpu_h3m_prepare_reconstruction(a->grid, psrc))) return ale_fail(c, "prepare_reconstruction (velocity)", rc);
    if ((rc = blomgpu_h3m_prepare_remapping(a->grid, a->map, pdst))) return ale_fail(c, "prepare_remapping (velocity)", rc);
    if ((rc = blomgpu_h3m_reconstruct(a->grid, a->vel, (isv ? h.f[F_v] : h.f[F_u]) + (size_t)nn * np))) return ale_fail(c, "reconstruct (velocity)", rc);
    if ((rc = blomgpu_h3m_remap(a->vel, a->map, rm))) return ale_fail(c, "remap (velocity)", rc);
    hipLaunchKernelGGL(k_ale_uv_back, gk, b, 0, c->stream, c->d, nn, isv, (const double *)rm);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
