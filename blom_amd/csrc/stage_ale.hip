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
// ... and with regrid_method = 'nudge', the reference's default (regrid_cntiso_hybrid_nudge_jslice, :560-916): the interfaces
// move a fraction delt1 / regrid_nudge_ts of the way towards their target densities, with a transition zone of adjusted
// targets (a quadratic Bezier curve, :680-728) below the pressure-level range, minimum thicknesses, a bound on the variation of
// neighbouring layer thicknesses (:848-913), and lateral smoothing of the interfaces where the stratification is weak
// (regrid_smooth_jslice, :946-1020).
// Neutral diffusion (ltedtp = 'neutral') is stage_ndiff.hip, called from here.  Not built: the z-level diagnostics (remap_trc_diazlv_jslice).
// Parity: cross-checked against the reference's REAL module compiled against a stand-in for mod_dia (oracle/Makefile
// *_xale, tests/test_xcheck_ale.py) -- a cross-check, not a pin.
#include "blomgpu_internal.h"
#include "eos.h"
#include "../../include/blomgpu_hor3map.h"

int h3m_use_stream(blomgpu_h3m_grid *G, hipStream_t stream);          // hor3map.hip
int h3m_sequence_begin(blomgpu_h3m_grid *G);
int h3m_sequence_begin_deferred(blomgpu_h3m_grid *G);
int h3m_sequence_end_deferred(blomgpu_h3m_grid *G);
int h3m_sequence_poll(blomgpu_h3m_grid *G, unsigned long long *column);
int h3m_sequence_end(blomgpu_h3m_grid *G);
int h3m_set_stream(blomgpu_h3m_grid *G, hipStream_t stream);
int h3m_set_active(blomgpu_h3m_grid *G, const int *active);
int h3m_extract_polycoeff_many(blomgpu_h3m_grid *G, int nf, blomgpu_h3m_src *const *srcs, double *const *outs);

#define H3M_MAXF 8
#include "stage_ndiff.h"

struct AleState {
  blomgpu_h3m_grid *grid = nullptr;
  std::vector<blomgpu_h3m_src *> trc;       // T, S, tracers (init_ale_regrid_remap :1412-1432)
  blomgpu_h3m_src *dens = nullptr;
  blomgpu_h3m_map *map = nullptr;
  // the u- and the v-columns side by side as ONE grid of 2 * nplane columns: every launch of the velocity part carries both
  // (the column routines are bound by the latency of a thread's own loads; twice the wavefronts hide twice as much of it)
  blomgpu_h3m_grid *grid_uv = nullptr;
  blomgpu_h3m_src *vel = nullptr;
  blomgpu_h3m_map *map_uv = nullptr;
  hipStream_t side = nullptr;               // launches that do not depend on each other run beside the model's stream
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // neutral diffusion runs on a stream of its own beside the remapping of the tracers AND the velocity part (which then has
  // its own scratch: plane_uv); ev_snap: the diffusion has taken its copy of pu, pv, which the velocity part overwrites
  hipStream_t side2 = nullptr;
  hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr, ev_snap = nullptr;
  double *plane_uv = nullptr;
  int *active = nullptr;                    // nplane + 2 * nplane flags: the columns the engine works on (the others are land or halo)
  double *plane = nullptr;                  // scratch: p_src, p_dst (kk+1 planes each), remapped fields (kk planes each)
  size_t plane_n = 0;
  int ntr_loc = 0, method = 0;
  // neutral diffusion (stage_ndiff.hip): polynomial coefficients of every field; T, S and drho/dT, drho/dS at the source
  // interfaces (8 kk planes), the flux convergence (kk ntr_loc planes) and the flux kernel's work arrays; ksmx, kdmx
  double *nd_tpc = nullptr, *nd_col = nullptr, *nd_rec = nullptr, *nd_recg = nullptr;
  size_t nd_nrec = 0;                 // record space per face of the neutral diffusion
  int *nd_ks = nullptr, *nd_reck = nullptr;
  int nd_npc = 0;
};

void ale_free(blomgpu_ctx *c) {
  AleState *a = (AleState *)c->ale;
  if (!a) return;
  if (a->grid) blomgpu_h3m_grid_free(a->grid);             // frees the sources and the map too
  if (a->grid_uv) blomgpu_h3m_grid_free(a->grid_uv);
  if (a->side) (void)hipStreamDestroy(a->side);
  if (a->ev_fork) (void)hipEventDestroy(a->ev_fork);
  if (a->ev_join) (void)hipEventDestroy(a->ev_join);
  if (a->side2) (void)hipStreamDestroy(a->side2);
  if (a->ev_fork2) (void)hipEventDestroy(a->ev_fork2);
  if (a->ev_join2) (void)hipEventDestroy(a->ev_join2);
  if (a->ev_snap) (void)hipEventDestroy(a->ev_snap);
  if (a->plane_uv) (void)hipFree(a->plane_uv);
  if (a->plane) (void)hipFree(a->plane);
  if (a->active) (void)hipFree(a->active);
  if (a->nd_tpc) (void)hipFree(a->nd_tpc);
  if (a->nd_col) (void)hipFree(a->nd_col);
  if (a->nd_ks) (void)hipFree(a->nd_ks);
  if (a->nd_rec) (void)hipFree(a->nd_rec);
  if (a->nd_recg) (void)hipFree(a->nd_recg);
  if (a->nd_reck) (void)hipFree(a->nd_reck);
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
  if ((rc = blomgpu_h3m_grid_create(&a->grid_uv, c->device, 2 * (int)h.nplane, h.kk, c->ale_method, c->ale_upper_bndr_ord, c->ale_lower_bndr_ord)))
    return ale_fail(c, "initialize_rcgs", rc);
  if ((rc = h3m_use_stream(a->grid_uv, c->stream))) return ale_fail(c, "stream", rc);
  (void)blomgpu_h3m_set_io(a->grid_uv, 2, 1);
  if ((rc = blomgpu_h3m_src_create(a->grid_uv, &a->vel, c->ale_velocity_limiting, c->ale_velocity_pc_upper, c->ale_velocity_pc_lower)))
    return ale_fail(c, "initialize_rcss", rc);
  if ((rc = blomgpu_h3m_map_create(a->grid_uv, &a->map_uv, h.kk))) return ale_fail(c, "initialize_rms", rc);
  HIPCHK(c, hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking));
  HIPCHK(c, hipEventCreateWithFlags(&a->ev_fork, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming));
  // potential density: density_limiting = 'monotonic' is the only value the reference accepts (:1287-1297)
  if ((rc = blomgpu_h3m_src_create(a->grid, &a->dens, BLOMGPU_H3M_MONOTONIC, c->ale_density_pc_upper, c->ale_density_pc_lower)))
    return ale_fail(c, "initialize_rcss", rc);
  if ((rc = blomgpu_h3m_map_create(a->grid, &a->map, h.kk))) return ale_fail(c, "initialize_rms", rc);
  // p_src, p_dst | remapped fields | polynomial coefficients of T and S (5 per layer at most) | sig_src | sig_trg
  a->plane_n = ((size_t)2 * (h.kk + 1) + (size_t)H3M_MAXF * h.kk + (size_t)10 * h.kk + h.kk + (h.kk + 1)) * h.nplane;
  HIPCHK(c, hipMalloc((void **)&a->plane, sizeof(double) * a->plane_n));
  HIPCHK(c, hipMalloc((void **)&a->active, sizeof(int) * 3 * h.nplane));
  if ((rc = h3m_set_active(a->grid, a->active)) || (rc = h3m_set_active(a->grid_uv, a->active + h.nplane))) return ale_fail(c, "active", rc);
  return 0;
}

static int ale_ndiff_buffers(blomgpu_ctx *c, AleState *a) {
  const DevView &h = c->h;
  const int npc = c->ale_method == BLOMGPU_H3M_PLM ? 2 : (c->ale_method == BLOMGPU_H3M_PPM ? 3 : 5);
  if (a->nd_tpc && a->nd_npc == npc) return 0;
  if (a->nd_tpc) { (void)hipFree(a->nd_tpc); a->nd_tpc = nullptr; }
  const size_t np = h.nplane, per = (size_t)h.kk * np;
  HIPCHK(c, hipMalloc((void **)&a->nd_tpc, sizeof(double) * (size_t)a->ntr_loc * npc * per));
  a->nd_npc = npc;
  if (!a->nd_col) {
    const size_t ncol = (size_t)(8 + a->ntr_loc) * per + ndiff_scratch_planes(h.kk) * 2 * np + (size_t)2 * (h.kk + 1) * np;
    // Record space per face.  Every neutral layer consumes a source or destination interface of one of the two columns, so 6 kk
    // records bound it (the default); that is (8 + ntr_loc) doubles + 2 ints per record and face, 2 nplane faces: 7.4 GB at channel
    // size with ntr = 3, 19 GB with 24 tracers.  The option ndiff_rec_per_face lowers it (a face that runs out raises the sticky
    // error word 4, nothing is silently truncated); the channel's faces use ~100 of the 318.
    const size_t nrec = c->ndiff_rec_per_face > 0 ? (size_t)c->ndiff_rec_per_face : (size_t)6 * h.kk;
    a->nd_nrec = nrec;
    {
      const double gb = (double)(sizeof(double) * nrec * (a->ntr_loc + 7) + sizeof(int) * (2 * nrec + 1)) * 2. * np / 1e9;
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) == hipSuccess && gb * 1e9 > (double)fr)
        return ctx_fail(c, "ndiff: the record space of ltedtp = 'neutral' needs " + std::to_string(gb) + " GB (6 kk records per face; lower it with the option ndiff_rec_per_face)");
    }
    HIPCHK(c, hipStreamCreateWithFlags(&a->side2, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&a->ev_fork2, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&a->ev_join2, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&a->ev_snap, hipEventDisableTiming));
    HIPCHK(c, hipMalloc((void **)&a->plane_uv, sizeof(double) * ((size_t)4 * (h.kk + 1) + (size_t)4 * h.kk) * np));
    HIPCHK(c, hipMalloc((void **)&a->nd_col, sizeof(double) * ncol));
    HIPCHK(c, hipMalloc((void **)&a->nd_ks, sizeof(int) * 2 * np));
    HIPCHK(c, hipMalloc((void **)&a->nd_rec, sizeof(double) * nrec * a->ntr_loc * 2 * np));
    HIPCHK(c, hipMalloc((void **)&a->nd_recg, sizeof(double) * nrec * 7 * 2 * np));
    HIPCHK(c, hipMalloc((void **)&a->nd_reck, sizeof(int) * (2 * nrec + 1) * 2 * np));
    HIPCHK(c, hipMemsetAsync(a->nd_col, 0, sizeof(double) * ncol, c->stream));
    HIPCHK(c, hipMemsetAsync(a->nd_ks, 0, sizeof(int) * 2 * np, c->stream));
    HIPCHK(c, hipMemsetAsync(a->nd_reck, 0, sizeof(int) * (2 * nrec + 1) * 2 * np, c->stream));
  }
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
                                double *__restrict__ pdst, int ring, int *__restrict__ active) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 - ring && j <= V.jj + ring && i >= 1 - ring && i <= V.ii + ring && V.m[I_ip][c];
  active[c] = col;                                               // the engine leaves the other columns out
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
                                 const double *__restrict__ pcS, int npc, double *__restrict__ sgs, double *__restrict__ sgt, int ring) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 - ring && j <= V.jj + ring && i >= 1 - ring && i <= V.ii + ring && V.m[I_ip][c];
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
  gd_t sigint = V.f[F_sigint];
  double sd2_prev = 0.;
  for (int k = 1; k <= ksmx; k++) {                                                   // :317-327
    const double sd1 = eos::sig(V.P, top(pcT, k), top(pcS, k));
    const double sd2 = eos::sig(V.P, bot(pcT, k), bot(pcS, k));
    PL(sigint, k) = k == 1 ? sd1 : .5 * (sd2_prev + sd1);
    sd2_prev = sd2;
  }
  for (int k = ksmx + 1; k <= kk; k++) PL(sigint, k) = sd2_prev;
  gcd_t sigma = V.f[F_sigma] + (size_t)nn * np, sigmar = V.f[F_sigmar];
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
                                  int k_range_plevel, double *__restrict__ pdst, int ring) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 - ring && j <= V.jj + ring && i >= 1 - ring && i <= V.ii + ring && V.m[I_ip][c];
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

// ---- vcoord_type = 'cntiso_hybrid', regrid_method = 'nudge': regrid_cntiso_hybrid_nudge_jslice, :560-916 -----------------------
struct NudgePar {
  double nudge_fac, stab_fac_limit, dpvar_fac, dpmin_interior;
  int k_range_plevel, dktzu, dktzl, ring;      // ring = 1: also the columns one ring beyond the tile (lateral smoothing follows)
};
__device__ inline double dsigdt_(const Params &P, double th, double s) {               // phy/mod_eos.F90:243-261
  const double r1 = P.ap11 + (P.ap12 + P.ap14 * th + P.ap15 * s) * th + (P.ap13 + P.ap16 * s) * s;
  const double r2i = 1. / (P.ap21 + (P.ap22 + P.ap24 * th + P.ap25 * s) * th + (P.ap23 + P.ap26 * s) * s);
  return (P.ap12 + 2. * P.ap14 * th + P.ap15 * s - (P.ap22 + 2. * P.ap24 * th + P.ap25 * s) * r1 * r2i) * r2i;
}
__device__ inline double dsigds_(const Params &P, double th, double s) {               // :306-323
  const double r1 = P.ap11 + (P.ap12 + P.ap14 * th + P.ap15 * s) * th + (P.ap13 + P.ap16 * s) * s;
  const double r2i = 1. / (P.ap21 + (P.ap22 + P.ap24 * th + P.ap25 * s) * th + (P.ap23 + P.ap26 * s) * s);
  return (P.ap13 + P.ap15 * th + 2. * P.ap16 * s - (P.ap23 + P.ap25 * th + 2. * P.ap26 * s) * r1 * r2i) * r2i;
}

// work planes (1-based level accessors below): sd1, sd2 = sig_srcdi(1,:), (2,:); sgt = sig_trg; dsg = dsig_trg; spm = sig_pmin;
// dpm = dpmin; sfac = smooth_fac
__global__ __launch_bounds__(256) void k_ale_nudge(const DevView *__restrict__ Vp, NudgePar Q, const double *__restrict__ psrc, const double *__restrict__ pcT,
                            const double *__restrict__ pcS, int npc, const double *__restrict__ plevel, double *__restrict__ sd1,
                            double *__restrict__ sd2, double *__restrict__ sgt, double *__restrict__ dsg, double *__restrict__ spm,
                            double *__restrict__ dpm, double *__restrict__ pdst, double *__restrict__ sfac) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk;
  const bool col = j >= 1 - Q.ring && j <= V.jj + Q.ring && i >= 1 - Q.ring && i <= V.ii + Q.ring && V.m[I_ip][c];
  if (!col) {
    for (int k = 1; k <= kk + 1; k++) { PL(pdst, k) = PL(psrc, k); PL(sfac, k) = 0.; }
    return;
  }
  auto pc_at = [&](const double *pc, int k, int q) { return q < npc ? pc[c + (size_t)(npc * (k - 1) + q) * np] : 0.; };
  // (peval0 / dpeval0 -- the polynomial and its derivative at a layer's top, coefficients 0 and 1 -- are not needed by this routine)
  auto bot = [&](const double *pc, int k) {                                                           // peval1
    double f = pc_at(pc, k, 0);
    for (int q = 1; q < 5; q++) f = f + pc_at(pc, k, q);
    return f;
  };
  auto dbot = [&](const double *pc, int k) {                                                          // dpeval1
    return pc_at(pc, k, 1) + 2. * pc_at(pc, k, 2) + 3. * pc_at(pc, k, 3) + 4. * pc_at(pc, k, 4);
  };
#define P(k) PL(psrc, k)
#define D(k) PL(pdst, k)
#define SD1(k) PL(sd1, k)
#define SD2(k) PL(sd2, k)
#define ST(k) PL(sgt, k)
#define DS(k) PL(dsg, k)
#define SP(k) PL(spm, k)
#define DM(k) PL(dpm, k)
#define SF(k) PL(sfac, k)
#define LEV(q) plevel[(q)-1]
  const double dpmin_interior = Q.dpmin_interior, nudge_fac = Q.nudge_fac, stab_fac_limit = Q.stab_fac_limit;
  // the plain per-level loops of the set-up with four (eight) levels' loads in flight and the level above in registers
  const double pbq = P(kk + 1);
  int ksmx = kk;                                                                     // :213-217
  for (int k0 = kk; k0 >= 1; k0 -= 8) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = P(k0 - u >= 1 ? k0 - u : 1);
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (k0 - u >= 1 && a[u] == pbq) ksmx = k0 - u - 1;
  }
  double sig_max = 0.;                                                                // :591-605
  gd_t sigint = V.f[F_sigint];
  {
    double sd2_prev = 0.;
    for (int k0 = 1; k0 <= ksmx; k0 += 4) {
      double cT[4][5], cS[4][5];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int kq = k0 + u <= ksmx ? k0 + u : ksmx;
#pragma unroll
        for (int q = 0; q < 5; q++) { cT[u][q] = pc_at(pcT, kq, q); cS[u][q] = pc_at(pcS, kq, q); }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u;
        if (k > ksmx) break;
        double tb = cT[u][0], sb = cS[u][0];                                          // peval1: the coefficients summed in order
#pragma unroll
        for (int q = 1; q < 5; q++) { tb = tb + cT[u][q]; sb = sb + cS[u][q]; }
        const double s1 = eos::sig(V.P, cT[u][0], cS[u][0]), s2 = eos::sig(V.P, tb, sb);
        SD1(k) = s1;
        SD2(k) = s2;
        sig_max = fmax2(sig_max, s2);
        PL(sigint, k) = k == 1 ? s1 : .5 * (sd2_prev + s1);
        sd2_prev = s2;
      }
    }
    for (int k = ksmx + 1; k <= kk; k++) PL(sigint, k) = sd2_prev;
  }
  gcd_t sigmar = V.f[F_sigmar];
  {                                                                                   // :608-615
    double st_prev = 0.;
    for (int k0 = 1; k0 <= kk; k0 += 8) {
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = PL(sigmar, k0 + u <= kk ? k0 + u : kk);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int k = k0 + u;
        if (k > kk) break;
        ST(k) = a[u];
        if (k >= 2) DS(k - 1) = a[u] - st_prev;
        st_prev = a[u];
      }
    }
    ST(kk + 1) = st_prev;
    DS(kk) = DS(kk - 1);
  }
  int kdmx;                                                                           // :620-623
  {
    int k = kk;
    for (; k >= 1; k--)
      if (ST(k) < sig_max) break;
    kdmx = k > 1 ? k : 1;
  }
  const double p1 = P(1), pb = P(kk + 1);
#define PMIN(k) fmin2(LEV(k) + p1, pb)
  int kl = 1;                                                                         // :635-652
  SP(1) = SD1(1);
  D(1) = PMIN(1);
  SF(1) = 1.;
  for (int k = 2; k <= Q.k_range_plevel; k++) {
    while (P(kl + 1) < PMIN(k)) kl = kl + 1;
    SP(k) = ((P(kl + 1) - PMIN(k)) * SD1(kl) + (PMIN(k) - P(kl)) * SD2(kl)) / (P(kl + 1) - P(kl));
    D(k) = P(k) + nudge_fac * (PMIN(k) - P(k));
    D(k) = fmin2(fmax2(fmax2(D(k), PMIN(k)), D(k - 1) + dpmin_interior), pb);
    SF(k) = 1.;
  }
  int kt = Q.k_range_plevel + 1;                                                       // :661-738
  while (kt <= kdmx) {
    while (P(kl + 1) < PMIN(kt)) kl = kl + 1;
    SP(kt) = ((P(kl + 1) - PMIN(kt)) * SD1(kl) + (PMIN(kt) - P(kl)) * SD2(kl)) / (P(kl + 1) - P(kl));
    if (ST(kt) > SP(kt)) {
      int ktzmin = kt - Q.dktzu;
      if (ktzmin < Q.k_range_plevel + 2) ktzmin = Q.k_range_plevel + 2;
      int ktzmax = kt + Q.dktzl;
      if (ktzmax > kk - 1) ktzmax = kk - 1;
      if (ktzmin < kt && ktzmax - ktzmin > 1) {
        const double ckt = (ST(kt) - SP(kt)) / (ST(kt) - ST(kt - 1) - SP(kt) + SP(kt - 1));
        const double sig_up = SP(ktzmin - 1) * ckt + SP(ktzmin) * (1. - ckt);
        const double sig_lo = ST(ktzmax - 1) * ckt + ST(ktzmax) * (1. - ckt);
        const double dk = (double)(ktzmax - ktzmin), dki = 1. / dk;
        double dsigdx_up = .5 * ((SP(ktzmin) - SP(ktzmin - 2)) * ckt + (SP(ktzmin + 1) - SP(ktzmin - 1)) * (1. - ckt)) * dk;
        const double dsigdx_lo = .5 * ((ST(ktzmax) - ST(ktzmax - 2)) * ckt + (ST(ktzmax + 1) - ST(ktzmax - 1)) * (1. - ckt)) * dk;
        dsigdx_up = fmax2(0., dsigdx_up);
        if (dsigdx_lo <= dsigdx_up || sig_up - sig_lo <= -dsigdx_lo || sig_up - sig_lo >= -dsigdx_up) {
          for (int k = ktzmin; k <= ktzmax - 1; k++) {
            const double x = ((double)(k - ktzmin) + ckt) * dki;
            ST(k) = sig_up * (1. - x) + sig_lo * x;
          }
        } else {
          const double xi = (sig_up - sig_lo + dsigdx_lo) / (dsigdx_lo - dsigdx_up);
          const double si = (dsigdx_lo * (sig_up + dsigdx_up) - dsigdx_up * sig_lo) / (dsigdx_lo - dsigdx_up);
          if (fabs(xi - .5) < 1.e-14) {
            for (int k = ktzmin; k <= ktzmax - 1; k++) {
              const double t = ((double)(k - ktzmin) + ckt) * dki;
              ST(k) = (1. - t) * ((1. - t) * sig_up + 2. * t * si) + t * t * sig_lo;
            }
          } else {
            for (int k = ktzmin; k <= ktzmax - 1; k++) {
              const double x = ((double)(k - ktzmin) + ckt) * dki;
              const double t = (sqrt(xi * (xi - 2. * x) + x) - xi) / (1. - 2. * xi);
              ST(k) = (1. - t) * ((1. - t) * sig_up + 2. * t * si) + t * t * sig_lo;
            }
          }
        }
        kt = ktzmin;
      }
      break;
    }
    D(kt) = P(kt) + nudge_fac * (PMIN(kt) - P(kt));
    D(kt) = fmin2(fmax2(fmax2(D(kt), PMIN(kt)), D(kt - 1) + dpmin_interior), pb);
    SF(kt) = 1.;
    kt = kt + 1;
  }
  for (int k = kt; k <= kk + 1; k++) { D(k) = pb; SF(k) = 0.; }                        // :744-747
  const int kend = ksmx < kdmx ? ksmx : kdmx;
  if (kt <= kend) {                                                                   // :749-818
    // four levels' coefficients, densities and pressures loaded ahead; what level k reads of level k-1 travels in registers
    double cTm[5], cSm[5];
#pragma unroll
    for (int q = 0; q < 5; q++) { cTm[q] = pc_at(pcT, kt - 1, q); cSm[q] = pc_at(pcS, kt - 1, q); }
    double sd2_m = SD2(kt - 1), ds_m = DS(kt - 1), p_m = P(kt - 1), p_0 = P(kt), d_m = D(kt - 1);
    for (int k0 = kt; k0 <= kend; k0 += 4) {
      double cT[4][5], cS[4][5], a_st[4], a_sd1[4], a_sd2[4], a_ds[4], a_p1[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int kq = k0 + u <= kend ? k0 + u : kend;
#pragma unroll
        for (int q = 0; q < 5; q++) { cT[u][q] = pc_at(pcT, kq, q); cS[u][q] = pc_at(pcS, kq, q); }
        a_st[u] = ST(kq); a_sd1[u] = SD1(kq); a_sd2[u] = SD2(kq); a_ds[u] = DS(kq); a_p1[u] = P(kq + 1);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u;
        if (k > kend) break;
        double stab_fac;
        double tu = cTm[0], su = cSm[0];                                              // peval1 of level k-1
#pragma unroll
        for (int q = 1; q < 5; q++) { tu = tu + cTm[q]; su = su + cSm[q]; }
        const double tl = cT[u][0], sl = cS[u][0];                                    // peval0 of level k
        const double dbT = cTm[1] + 2. * cTm[2] + 3. * cTm[3] + 4. * cTm[4], dbS = cSm[1] + 2. * cSm[2] + 3. * cSm[3] + 4. * cSm[4];
        const double dtT = cT[u][1], dtS = cS[u][1];
        const double st = a_st[u], sd1 = a_sd1[u], p_1 = a_p1[u], ds_0 = a_ds[u];
        double dn;
        if (st < sd2_m && st < sd1) {
          const double dsig = st - sd2_m;
          double dsigdx = dsigdt_(V.P, tu, su) * dbT + dsigds_(V.P, tu, su) * dbS;
          stab_fac = dsigdx / ds_m;
          dsigdx = ds_m * fmax2(stab_fac, stab_fac_limit);
          dn = p_0 + fmax2(-.5, dsig * nudge_fac / dsigdx) * (p_0 - p_m);
        } else if (st > sd2_m && st > sd1) {
          const double dsig = st - sd1;
          double dsigdx = dsigdt_(V.P, tl, sl) * dtT + dsigds_(V.P, tl, sl) * dtS;
          stab_fac = dsigdx / ds_0;
          dsigdx = ds_0 * fmax2(stab_fac, stab_fac_limit);
          dn = p_0 + fmin2(.5, dsig * nudge_fac / dsigdx) * (p_1 - p_0);
        } else {
          const double dsigdx_up = dsigdt_(V.P, tu, su) * dbT + dsigds_(V.P, tu, su) * dbS;
          const double dsigdx_lo = dsigdt_(V.P, tl, sl) * dtT + dsigds_(V.P, tl, sl) * dtS;
          const double dp_up = fmax2(p_0 - p_m, ALE_EPSILP), dp_lo = fmax2(p_1 - p_0, ALE_EPSILP);
          double sig_intrp = ((sd1 + .5 * dsigdx_lo) * dp_up + (sd2_m - .5 * dsigdx_up) * dp_lo) / (dp_up + dp_lo);
          sig_intrp = fmax2(fmin2(sd2_m, sd1), fmin2(fmax2(sd2_m, sd1), sig_intrp));
          const double dsig = st - sig_intrp;
          if (dsig < 0.) {
            double dsigdx = dsigdx_up + 2. * (sig_intrp - sd2_m);
            stab_fac = dsigdx / ds_m;
            dsigdx = ds_m * fmax2(stab_fac, stab_fac_limit);
            dn = p_0 + fmax2(-.5, dsig * nudge_fac / dsigdx) * (p_0 - p_m);
          } else {
            double dsigdx = dsigdx_lo + 2. * (sd1 - sig_intrp);
            stab_fac = dsigdx / ds_0;
            dsigdx = ds_0 * fmax2(stab_fac, stab_fac_limit);
            dn = p_0 + fmin2(.5, dsig * nudge_fac / dsigdx) * (p_1 - p_0);
          }
        }
        dn = fmin2(fmax2(fmax2(dn, PMIN(k)), d_m + dpmin_interior), pb);
        D(k) = dn;
        SF(k) = fmax2(0., fmin2(1., (stab_fac_limit - stab_fac) / stab_fac_limit));
#pragma unroll
        for (int q = 0; q < 5; q++) { cTm[q] = cT[u][q]; cSm[q] = cS[u][q]; }
        sd2_m = a_sd2[u]; ds_m = ds_0; p_m = p_0; p_0 = p_1; d_m = dn;
      }
    }
  }
  for (int k = (kt > kend ? kt : kend) + 1; k <= kdmx; k++) {                           // :820-841
    if (ST(k) < SD2(ksmx)) {
      const double tb = bot(pcT, ksmx), sb = bot(pcS, ksmx);
      const double dsig = ST(k) - SD2(ksmx);
      double dsigdx = dsigdt_(V.P, tb, sb) * dbot(pcT, ksmx) + dsigds_(V.P, tb, sb) * dbot(pcS, ksmx);
      const double stab_fac = dsigdx / DS(ksmx - 1);
      dsigdx = DS(ksmx - 1) * fmax2(stab_fac, stab_fac_limit);
      D(k) = pb + fmax2(-.5, dsig * nudge_fac / dsigdx) * (pb - P(ksmx));
      D(k) = fmin2(fmax2(fmax2(D(k), PMIN(k)), D(k - 1) + dpmin_interior), pb);
      SF(k) = fmax2(0., fmin2(1., (stab_fac_limit - stab_fac) / stab_fac_limit));
    }
  }
  // limit the local variation of the layer thicknesses, :848-913
  int ks = kt, ke = kk;
  for (int k = kk; k >= 1; k--)
    if (D(k) == D(kk + 1)) ke = k - 1;
  for (int k = ks; k <= ke - 1; k++)
    DM(k) = fmin2(2. * D(ke + 1) - D(k + 1) - D(k), fmax2(dpmin_interior, Q.dpvar_fac * (D(k + 2) - D(k - 1)) / 3.));
  int k = ks;
  while (k < ke) {
    if (D(k + 1) - D(k) < DM(k)) {
      int ku = k;
      kl = k + 1;
      double dpmin_sum = DM(ku), dp_sum = D(k + 1) - D(k);
      bool added = true;
      while (added) {
        added = false;
        if (kl + 1 < ke) {
          const double a = dpmin_sum + DM(kl), b = dp_sum + D(kl + 1) - D(kl);
          if (a > b) { dpmin_sum = a; dp_sum = b; kl = kl + 1; added = true; }
        }
        if (ku > ks) {
          const double a = dpmin_sum + DM(ku - 1), b = dp_sum + D(ku) - D(ku - 1);
          if (a > b) { dpmin_sum = a; dp_sum = b; ku = ku - 1; added = true; }
        }
      }
      if (ku == ks) {
        for (int q = ks; q <= kl - 1; q++) D(q + 1) = fmin2(D(ke + 1), D(q) + DM(q));
        for (int q = kl; q <= ke - 1; q++) D(q + 1) = fmax2(D(q + 1), D(q));
        ks = kl - 1;
        k = ks;
      } else {
        const double dp_up = D(ku) - D(ku - 1), dp_lo = D(kl + 1) - D(kl);
        D(ku) = fmax2(D(ku - 1), D(ku) - (dpmin_sum - dp_sum) * dp_up / fmax2(ALE_EPSILP, dp_up + dp_lo));
        for (int q = ku; q <= kl - 1; q++) D(q + 1) = fmin2(D(ke + 1), D(q) + DM(q));
        k = kl;
      }
    }
    k = k + 1;
  }
#undef PMIN
#undef LEV
#undef SF
#undef DM
#undef SP
#undef DS
#undef ST
#undef SD2
#undef SD1
#undef D
#undef P
}

// regrid_smooth_jslice, :946-1020: lateral smoothing of the regridded interfaces where the stratification is weak.  The
// reference accumulates the flux convergence of a cell while it walks the faces of three rows at a time: for the cell (i,j) in
// the order - u-face i, + u-face i+1, - v-face j, + v-face j+1 (faces that are no velocity points contribute nothing); all
// fluxes are formed from the unsmoothed interfaces.  Here one thread per cell and interface gathers them in that order.
__global__ void k_ale_smooth(const DevView *__restrict__ Vp, double smooth_diff_max, const double *__restrict__ pdst,
                             const double *__restrict__ sfac, double *__restrict__ pout, int ring) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const size_t np = V.nplane;
  const int kk = V.kk, ni = V.ni;
  const int k = blockIdx.y + 1;                                  // interface 1..kk+1
  double v = pdst[c + (size_t)(k - 1) * np];
  // ring = 1 with neutral diffusion, which reads the smoothed interfaces of the cells one ring beyond the tile (:1629-1634)
  const bool cell = j >= 1 - ring && j <= V.jj + ring && i >= 1 - ring && i <= V.ii + ring && V.m[I_ip][c];
  if (cell && k >= 2 && k <= kk) {
    const double delt1 = V.P.delt1;
    gcd_t scp2 = V.f[F_scp2], difmxp = V.f[F_difmxp];
#define D(x, q) pdst[(x) + (size_t)((q)-1) * np]
#define SFC(x, q) sfac[(x) + (size_t)((q)-1) * np]
    auto uflux = [&](size_t cc) {                                // u-face between cc-1 and cc, :962-984
      const size_t w = cc - 1;
      const double cdiff = delt1 * V.f[F_scuy][cc] * V.f[F_scuxi][cc];
      const double difmx = .5 * (difmxp[w] + difmxp[cc]);
      const double flxhi = .125 * fmin2((D(w, k) - D(w, k - 1)) * scp2[w], (D(cc, k + 1) - D(cc, k)) * scp2[cc]);
      const double flxlo = -.125 * fmin2((D(cc, k) - D(cc, k - 1)) * scp2[cc], (D(w, k + 1) - D(w, k)) * scp2[w]);
      const double sdiff = fmin2(.5 * (SFC(w, k) + SFC(cc, k)) * smooth_diff_max, difmx);
      return fmin2(flxhi, fmax2(flxlo, cdiff * sdiff * (D(w, k) - D(cc, k))));
    };
    auto vflux = [&](size_t cc) {                                // v-face between cc-ni and cc, :986-1008
      const size_t sdn = cc - ni;
      const double cdiff = delt1 * V.f[F_scvx][cc] * V.f[F_scvyi][cc];
      const double difmx = .5 * (difmxp[sdn] + difmxp[cc]);
      const double flxhi = .125 * fmin2((D(sdn, k) - D(sdn, k - 1)) * scp2[sdn], (D(cc, k + 1) - D(cc, k)) * scp2[cc]);
      const double flxlo = -.125 * fmin2((D(cc, k) - D(cc, k - 1)) * scp2[cc], (D(sdn, k + 1) - D(sdn, k)) * scp2[sdn]);
      const double sdiff = fmin2(.5 * (SFC(sdn, k) + SFC(cc, k)) * smooth_diff_max, difmx);
      return fmin2(flxhi, fmax2(flxlo, cdiff * sdiff * (D(sdn, k) - D(cc, k))));
    };
    double conv = 0.;
    if (V.m[I_iu][c]) conv = conv - uflux(c);
    if (V.m[I_iu][c + 1]) conv = conv + uflux(c + 1);
    if (V.m[I_iv][c]) conv = conv - vflux(c);
    if (V.m[I_iv][c + ni]) conv = conv + vflux(c + ni);
    v = v - conv * V.f[F_scp2i][c];
#undef SFC
#undef D
  }
  pout[c + (size_t)(k - 1) * np] = v;
}

// copy_jslice_to_3d (:1153-1179) for up to H3M_MAXF remapped fields starting with field f0 of (T, S, tracer 1, ..)
// flx != nullptr: ndiff_update_trc_jslice (phy/mod_ndiff.F90:1149-1175) first -- the flux convergence [layer][field] of the
// neutral diffusion comes off the remapped value
__global__ void k_ale_copy_back(const DevView *__restrict__ Vp, int nn, const double *__restrict__ pdst, const double *__restrict__ rm,
                                int f0, int nf, const double *__restrict__ flx, int ntr_loc) {
  const DevView &V = *Vp;
  PLANE_T(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane, per = (size_t)V.kk * np;
  const int k = blockIdx.y;
  const size_t okn = c + (size_t)(k + nn) * np, ok = c + (size_t)k * np;
  double q = 0.;
  if (flx) q = 1. / (V.f[F_scp2][c] * fmax2(pdst[c + (size_t)(k + 1) * np] - pdst[ok], 1.e-5));
  double ts[2] = {0., 0.};
  for (int f = 0; f < nf; f++) {
    const int nt = f0 + f;                                 // 0: T, 1: S, 2..: tracers
    double v = rm[(size_t)f * per + ok];
    if (flx) v = v - q * flx[c + ((size_t)k * ntr_loc + nt) * np];
    if (nt == 0) { V.f[F_temp][okn] = v; ts[0] = v; }
    else if (nt == 1) { V.f[F_saln][okn] = v; ts[1] = v; }
    else V.f[F_trc][okn + (size_t)(nt - 2) * 2 * V.kk * np] = v;
  }
  if (f0 == 0) {                                           // T and S travel in the first group: dp and sigma with them
    V.f[F_dp][okn] = pdst[c + (size_t)(k + 1) * np] - pdst[ok];
    V.f[F_sigma][okn] = eos::sig(V.P, ts[0], ts[1]);
  }
}

// the new layer thicknesses alone (copy_jslice_to_3d's dp, :1168), ahead of the copy-back of the tracers
__global__ void k_ale_dp_new(const DevView *__restrict__ Vp, int nn, const double *__restrict__ pdst) {
  const DevView &V = *Vp;
  PLANE_T(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const size_t np = V.nplane;
  const int k = blockIdx.y;
  V.f[F_dp][c + (size_t)(k + nn) * np] = pdst[c + (size_t)(k + 1) * np] - pdst[c + (size_t)k * np];
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

// source and destination interfaces of the u- (blockIdx.y = 0) and the v-columns (1) (:1768-1779, :1836-1847), and the
// velocities themselves, as columns c and nplane + c of one grid of 2 * nplane columns: the source interfaces are the old ones
// rescaled to the new depth of the column
__global__ void k_ale_uv_src_dst(const DevView *__restrict__ Vp, int nn, double *__restrict__ psrc, double *__restrict__ pdst,
                                 double *__restrict__ uin, int *__restrict__ active) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const int isv = blockIdx.y;
  const size_t np = V.nplane, nc = 2 * np, cc = c + (size_t)isv * np;
  const int kk = V.kk;
  const bool col = j >= 1 && j <= V.jj && i >= 1 && i <= V.ii && (isv ? V.m[I_iv][c] : V.m[I_iu][c]);
  active[cc] = col;
  if (!col) return;                                              // the engine leaves the column out; nothing reads its planes
  gcd_t pz = isv ? V.f[F_pv] : V.f[F_pu], dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gcd_t uz = (isv ? V.f[F_v] : V.f[F_u]) + (size_t)nn * np;
  double a = pz[c];
  pdst[cc] = a;
  for (int k = 0; k < kk; k++) { a = a + dpz[c + (size_t)k * np]; pdst[cc + (size_t)(k + 1) * nc] = a; }
  gcd_t u1 = V.f[F_util1];
  const double q = fmin2(u1[isv ? c - V.ni : c - 1], u1[c]) / pz[c + (size_t)kk * np];
  for (int k = 0; k <= kk; k++) psrc[cc + (size_t)k * nc] = pz[c + (size_t)k * np] * q;
  for (int k = 0; k < kk; k++) uin[cc + (size_t)k * nc] = uz[c + (size_t)k * np];
}

__global__ void k_ale_uv_back(const DevView *__restrict__ Vp, int nn, const double *__restrict__ rm) {
  const DevView &V = *Vp;
  PLANE_T(V);
  const int isv = blockIdx.z;
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const int k = blockIdx.y;
  (isv ? V.f[F_v] : V.f[F_u])[c + (size_t)(k + nn) * V.nplane] = rm[c + (size_t)isv * V.nplane + (size_t)k * 2 * V.nplane];
}

int st_ale_regrid_remap(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)n; (void)mm; (void)k1m;
  const DevView &h = c->h;
  if (h.P.vcoord_tag == 1) return ctx_fail(c, "ale_regrid_remap: vcoord_type = 'isopyc_bulkml' has no ALE step (phy/mod_blom_step.F90:138-149)");
  if (h.P.vcoord_tag != 2 && h.P.vcoord_tag != 3) return ctx_fail(c, "ale_regrid_remap: unknown vertical coordinate");
  if (h.P.ltedtp_opt != 1 && h.P.ltedtp_opt != 2) return ctx_fail(c, "ale_regrid_remap: unknown ltedtp_opt");
  if (!c->ale_plevel) return ctx_fail(c, "ale_regrid_remap: the pressure levels are not set (blomgpu_set_vector \"plevel\", phy/mod_vcoord.F90:99)");
  if (int rc = ale_prepare(c)) return rc;
  TimeScope ts(c, "ale_regrid_remap");
  AleState *a = (AleState *)c->ale;
  const size_t np = h.nplane, per = (size_t)h.kk * np;
  double *psrc = a->plane, *pdst = psrc + (size_t)(h.kk + 1) * np, *rm = pdst + (size_t)(h.kk + 1) * np;
  const dim3 g1((unsigned)((np + 255) / 256)), gk((unsigned)((np + 255) / 256), h.kk), b(256);
  int rc;
  // the status of the engine's calls is read back once, at the end -- or, inside blomgpu_step, with the other sticky error
  // words every check_period steps (ale_check_deferred): no host synchronisation inside the step
  const bool defer = c->defer_checks;
  // an error return between here and the end of the stage must leave no work of the second diffusion stream unjoined and the
  // engine's grids out of their sticky mode (the status then reaches the caller through the failing call itself)
  struct Unwind {
    AleState *a; bool armed = true, forked2 = false;
    ~Unwind() {
      if (!armed) return;
      if (forked2 && a->side2) (void)hipStreamSynchronize(a->side2);
      (void)h3m_sequence_end_deferred(a->grid);
      (void)h3m_sequence_end_deferred(a->grid_uv);
    }
  } unwind{a};
  if (defer) {
    if ((rc = h3m_sequence_begin_deferred(a->grid)) || (rc = h3m_sequence_begin_deferred(a->grid_uv))) return ale_fail(c, "sequence", rc);
  } else if ((rc = h3m_sequence_begin(a->grid)) || (rc = h3m_sequence_begin(a->grid_uv))) return ale_fail(c, "sequence", rc);
  // ---- tracers ----------------------------------------------------------------------------------------------------------
  const bool nudge = h.P.vcoord_tag == 2 && c->ale_regrid_method == 2;
  const bool smooth = nudge && c->ale_smooth_diff_max > 0.;
  const bool ndiff = h.P.ltedtp_opt == 2;
  // the j-slice offsets of :1559-1569: neutral diffusion reads the regridded columns one ring beyond the tile, the lateral
  // smoothing those one ring beyond the cells it smooths
  const int jofs2 = ndiff ? 1 : 0, ring = jofs2 + (smooth ? 1 : 0);
  if (ring) {                                                          // :1603-1607
    if (int rc2 = st_xctilr(c, h.f[F_temp] + (size_t)(k1n - 1) * np, 1, h.kk, ring, ring, 1)) return rc2;
    if (int rc2 = st_xctilr(c, h.f[F_saln] + (size_t)(k1n - 1) * np, 1, h.kk, ring, ring, 1)) return rc2;
    if (int rc2 = st_xctilr(c, h.f[F_sigma] + (size_t)(k1n - 1) * np, 1, h.kk, ring, ring, 1)) return rc2;
  }
  if (ndiff) {                                                         // :1608-1613
    for (int nt = 0; nt < h.ntr; nt++)
      if (int rc2 = st_xctilr(c, h.f[F_trc] + ((size_t)(k1n - 1) + (size_t)nt * 2 * h.kk) * np, 1, h.kk, 1, 1, 1)) return rc2;
    if (c->ndiff_surface_align)
      if (int rc2 = st_xctilr(c, h.f[F_dpml], 1, 1, 1, 1, 1)) return rc2;
    if (int rc2 = ale_ndiff_buffers(c, a)) return rc2;
  }
  hipLaunchKernelGGL(k_ale_p_src_dst, g1, b, 0, c->stream, c->d, nn, (const double *)c->ale_plevel, psrc, pdst, ring, a->active);
  if ((rc = blomgpu_h3m_prepare_reconstruction(a->grid, psrc))) return ale_fail(c, "prepare_reconstruction", rc);
  // Beside the model's stream: the reconstruction of the first batch's fields that the regridding does not look at -- all of
  // them for pressure levels, the tracers for the density-following coordinate (T and S are reconstructed for the regridding
  // and, the grid and the data being the same, not a second time for the remapping).
  auto field = [&](int nt) -> const double * {
    return nt == 0 ? h.f[F_temp] + (size_t)nn * np : nt == 1 ? h.f[F_saln] + (size_t)nn * np : h.f[F_trc] + ((size_t)nn + (size_t)(nt - 2) * 2 * h.kk) * np;
  };
  const int npc = c->ale_method == BLOMGPU_H3M_PLM ? 2 : (c->ale_method == BLOMGPU_H3M_PPM ? 3 : 5);
  const int nf0 = a->ntr_loc < H3M_MAXF ? a->ntr_loc : H3M_MAXF, lo0 = h.P.vcoord_tag == 2 ? 2 : 0;
  auto fork = [&]() -> int {
    HIPCHK(c, hipEventRecord(a->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(a->side, a->ev_fork, 0));
    return 0;
  };
  auto join = [&]() -> int {
    HIPCHK(c, hipEventRecord(a->ev_join, a->side));
    HIPCHK(c, hipStreamWaitEvent(c->stream, a->ev_join, 0));
    return 0;
  };
  // with neutral diffusion the coefficients of every field are kept ([field][layer][coefficient] planes, a->nd_tpc), and every
  // field is reconstructed before the fluxes are formed
  auto coeff = [&](int nt) -> double * { return a->nd_tpc + (size_t)nt * npc * per; };
  if (nf0 > lo0) {
    const double *us[H3M_MAXF];
    blomgpu_h3m_src *ss[H3M_MAXF];
    double *pcs[H3M_MAXF];
    for (int f = lo0; f < nf0; f++) { us[f - lo0] = field(f); ss[f - lo0] = a->trc[f]; pcs[f - lo0] = ndiff ? coeff(f) : nullptr; }
    if (int rc2 = fork()) return rc2;
    (void)h3m_set_stream(a->grid, a->side);
    rc = blomgpu_h3m_reconstruct_many(a->grid, nf0 - lo0, ss, us);
    if (!rc && ndiff) rc = h3m_extract_polycoeff_many(a->grid, nf0 - lo0, ss, pcs);
    (void)h3m_set_stream(a->grid, c->stream);
    if (rc) return ale_fail(c, "reconstruct", rc);
  }
  if (ndiff)
    for (int f0 = H3M_MAXF; f0 < a->ntr_loc; f0 += H3M_MAXF) {
      const int nf = a->ntr_loc - f0 < H3M_MAXF ? a->ntr_loc - f0 : H3M_MAXF;
      const double *us[H3M_MAXF];
      blomgpu_h3m_src *ss[H3M_MAXF];
      double *pcs[H3M_MAXF];
      for (int f = 0; f < nf; f++) { us[f] = field(f0 + f); ss[f] = a->trc[f0 + f]; pcs[f] = coeff(f0 + f); }
      if ((rc = blomgpu_h3m_reconstruct_many(a->grid, nf, ss, us))) return ale_fail(c, "reconstruct", rc);
      if ((rc = h3m_extract_polycoeff_many(a->grid, nf, ss, pcs))) return ale_fail(c, "extract_polycoeff", rc);
    }
  if (h.P.vcoord_tag == 2) {
    // regrid_cntiso_hybrid_direct_jslice: the interfaces go where the reconstructed potential density takes its target values
    double *pcT = rm + (size_t)H3M_MAXF * per, *pcS = pcT + (size_t)5 * per, *sgs = pcS + (size_t)5 * per, *sgt = sgs + per;
    if (ndiff) { pcT = coeff(0); pcS = coeff(1); }
    const double *ts[2] = {field(0), field(1)};
    blomgpu_h3m_src *tss[2] = {a->trc[0], a->trc[1]};
    double *pcs[2] = {pcT, pcS};
    if ((rc = blomgpu_h3m_reconstruct_many(a->grid, 2, tss, ts))) return ale_fail(c, "reconstruct", rc);
    if ((rc = h3m_extract_polycoeff_many(a->grid, 2, tss, pcs))) return ale_fail(c, "extract_polycoeff", rc);
    if (nudge) {
      // regrid_cntiso_hybrid_nudge_jslice; its work arrays lie where the remapped fields will (nothing has been remapped yet)
      double *sd1 = rm, *sd2 = rm + per, *dsg = rm + 2 * per, *dpm = rm + 3 * per, *spm = rm + 4 * per, *sfac = rm + 6 * per;
      NudgePar Q;
      Q.nudge_fac = h.P.delt1 / c->ale_regrid_nudge_ts;                                       // :632
      Q.stab_fac_limit = c->ale_stab_fac_limit; Q.dpvar_fac = c->ale_dpvar_fac; Q.dpmin_interior = c->ale_dpmin_interior;
      Q.k_range_plevel = c->ale_k_range_plevel; Q.dktzu = c->ale_dktzu; Q.dktzl = c->ale_dktzl; Q.ring = ring;
      hipLaunchKernelGGL(k_ale_nudge, g1, b, 0, c->stream, c->d, Q, (const double *)psrc, (const double *)pcT, (const double *)pcS, npc,
                         (const double *)c->ale_plevel, sd1, sd2, sgt, dsg, spm, dpm, pdst, sfac);
      if (smooth) {                                                                           // regrid_smooth_jslice
        double *pout = sgs;                               // kk+1 planes: sgs and the first plane of sgt, both free again
        hipLaunchKernelGGL(k_ale_smooth, dim3((unsigned)((np + 255) / 256), h.kk + 1), b, 0, c->stream, c->d, c->ale_smooth_diff_max,
                           (const double *)pdst, (const double *)sfac, pout, jofs2);
        pdst = pout;                                      // the smoothed interfaces are the destination grid from here on
      }
    } else {
    hipLaunchKernelGGL(k_ale_direct_pre, g1, b, 0, c->stream, c->d, nn, (const double *)psrc, (const double *)pcT, (const double *)pcS,
                       npc, sgs, sgt, ring);
    if ((rc = blomgpu_h3m_reconstruct(a->grid, a->dens, sgs))) return ale_fail(c, "reconstruct (density)", rc);
    if ((rc = blomgpu_h3m_regrid(a->dens, h.kk + 1, sgt, pdst, -1.e33, 0))) return ale_fail(c, "regrid", rc);
    hipLaunchKernelGGL(k_ale_direct_post, g1, b, 0, c->stream, c->d, (const double *)psrc, (const double *)sgs, (const double *)sgt,
                       (const double *)c->ale_plevel, c->ale_dpmin_interior, c->ale_k_range_plevel, pdst, ring);
    }
  }
  if (nf0 > lo0)
    if (int rc2 = join()) return rc2;
  if (ndiff) {
    // ndiff_prep_jslice, ndiff_uflx_jslice, ndiff_vflx_jslice (:1639-1664): the flux convergence of every destination layer.
    // Beside the model's stream: the searches wait on their own loads most of the time, the remapping of the first batch of
    // fields does not need their result (the copy-back does).
    NdArgs A;
    A.psrc = psrc; A.pdst = pdst; A.ksmx = a->nd_ks; A.kdmx = a->nd_ks + np; A.tpc = a->nd_tpc; A.tsd = a->nd_col;
    A.drt = a->nd_col + (size_t)4 * per; A.drs = a->nd_col + (size_t)6 * per; A.flx = a->nd_col + (size_t)8 * per;
    A.scr = A.flx + (size_t)a->ntr_loc * per;
    A.rec_n = a->nd_reck; A.rec_k = a->nd_reck + 2 * np; A.rec_s = A.rec_k + a->nd_nrec * 2 * np; A.rec_g = a->nd_recg;
    A.rec_f = a->nd_rec; A.nrec_max = (int)a->nd_nrec;
    A.prof = c->kprof; A.prof_words = c->kprof_words;     // (its own buffer with its size: until round 6 the barotropic profiler's, unchecked)
    A.flux_zero = c->in_sequence && c->fluxes_zeroed ? 1 : 0;
    A.puv = A.scr + ndiff_scratch_planes(h.kk) * 2 * np;
    A.kk = h.kk; A.npc = npc; A.ntr_loc = a->ntr_loc; A.mm = mm; A.nn = nn; A.surface_align = c->ndiff_surface_align;
    HIPCHK(c, hipEventRecord(a->ev_fork2, c->stream));
    HIPCHK(c, hipStreamWaitEvent(a->side2, a->ev_fork2, 0));
    unwind.forked2 = true;
    if (int rc2 = st_ndiff_prep_flux(c, a->side2, a->ev_snap, A, a->nd_ks, a->nd_ks + np, a->nd_col, a->nd_col + (size_t)4 * per, a->nd_col + (size_t)6 * per))
      return rc2;
    c->fluxes_zeroed = false;                   // utflx .. vsflx carry the diffusive fluxes now: advect must add to them
  }
  if ((rc = blomgpu_h3m_prepare_remapping(a->grid, a->map, pdst))) return ale_fail(c, "prepare_remapping", rc);
  const double *flx = ndiff ? a->nd_col + (size_t)8 * per : nullptr;
  // ---- velocities, :1692-1900 -------------------------------------------------------------------------------------------------
  // uvp: their scratch planes (where the tracers' scratch was, unless the neutral diffusion is still reading that)
  auto velocities = [&](double *uvp) -> int {
    hipLaunchKernelGGL(k_ale_pupv, g1, b, 0, c->stream, c->d, nn);
    if (int rc2 = st_xctilr(c, h.f[F_dp] + (size_t)(k1n - 1) * np, 1, h.kk, 3, 3, 1)) return rc2;
    hipLaunchKernelGGL(k_ale_pscan, g1, b, 0, c->stream, c->d, nn);
    // dpu, dpv of the new layers and their copies dpuold, dpvold, j,i = -1..+2 (:1735-1762)
    if (int rc2 = launch_dpudpv(c, nn, 4)) return rc2;
    // u- and v-columns as one grid of 2 * nplane columns
    double *ps2 = uvp, *pd2 = ps2 + (size_t)2 * (h.kk + 1) * np, *ui2 = pd2 + (size_t)2 * (h.kk + 1) * np, *rm2 = ui2 + (size_t)2 * per;
    hipLaunchKernelGGL(k_ale_uv_src_dst, dim3((unsigned)((np + 255) / 256), 2), b, 0, c->stream, c->d, nn, ps2, pd2, ui2, a->active + np);
    if ((rc = blomgpu_h3m_prepare_reconstruction(a->grid_uv, ps2))) return ale_fail(c, "prepare_reconstruction (velocity)", rc);
    // the segments of the destination grid and the reconstruction need the prepared grid, not each other
    if (int rc2 = fork()) return rc2;
    (void)h3m_set_stream(a->grid_uv, a->side);
    rc = blomgpu_h3m_reconstruct(a->grid_uv, a->vel, ui2);
    (void)h3m_set_stream(a->grid_uv, c->stream);
    if (rc) return ale_fail(c, "reconstruct (velocity)", rc);
    if ((rc = blomgpu_h3m_prepare_remapping(a->grid_uv, a->map_uv, pd2))) return ale_fail(c, "prepare_remapping (velocity)", rc);
    if (int rc2 = join()) return rc2;
    if ((rc = blomgpu_h3m_remap(a->vel, a->map_uv, rm2))) return ale_fail(c, "remap (velocity)", rc);
    hipLaunchKernelGGL(k_ale_uv_back, dim3((unsigned)((np + 255) / 256), h.kk, 2), b, 0, c->stream, c->d, nn, (const double *)rm2);
    return 0;
  };
  bool vel_done = false;
  for (int f0 = 0; f0 < a->ntr_loc; f0 += H3M_MAXF) {
    const int nf = a->ntr_loc - f0 < H3M_MAXF ? a->ntr_loc - f0 : H3M_MAXF;
    const double *us[H3M_MAXF];
    double *ud[H3M_MAXF];
    blomgpu_h3m_src *ss[H3M_MAXF];
    for (int f = 0; f < nf; f++) {
      us[f] = field(f0 + f);
      ud[f] = rm + (size_t)f * per;
      ss[f] = a->trc[f0 + f];
    }
    if (f0 > 0 && !ndiff && (rc = blomgpu_h3m_reconstruct_many(a->grid, nf, ss, us))) return ale_fail(c, "reconstruct", rc);
    if ((rc = blomgpu_h3m_remap_many(nf, ss, a->map, ud))) return ale_fail(c, "remap", rc);
    if (ndiff && f0 == 0) {
      // While the neutral diffusion is at work beside: the new layer thicknesses (all the velocity part needs of the tracers'
      // result) and the velocity part itself, once the diffusion has its copy of pu, pv.  Then wait for the flux convergence.
      hipLaunchKernelGGL(k_ale_dp_new, gk, b, 0, c->stream, c->d, nn, (const double *)pdst);
      HIPCHK(c, hipStreamWaitEvent(c->stream, a->ev_snap, 0));
      if (int rc2 = velocities(a->plane_uv)) return rc2;
      vel_done = true;
      HIPCHK(c, hipEventRecord(a->ev_join2, a->side2));
      HIPCHK(c, hipStreamWaitEvent(c->stream, a->ev_join2, 0));
      unwind.forked2 = false;
    }
    hipLaunchKernelGGL(k_ale_copy_back, gk, b, 0, c->stream, c->d, nn, (const double *)pdst, (const double *)rm, f0, nf, flx, a->ntr_loc);
  }
  if (!vel_done)
    if (int rc2 = velocities(a->plane)) return rc2;
  HIPCHK(c, hipGetLastError());
  unwind.armed = false;
  if (defer) {
    if ((rc = h3m_sequence_end_deferred(a->grid)) || (rc = h3m_sequence_end_deferred(a->grid_uv))) return ale_fail(c, "sequence", rc);
    return 0;
  }
  if ((rc = h3m_sequence_end(a->grid)) || (rc = h3m_sequence_end(a->grid_uv))) return ale_fail(c, "a column failed", rc);
  return 0;
}

// the deferred status of ale_regrid_remap's engine calls (blomgpu_step): 0, or the failure of a column in one of the steps
// since the last check
int ale_check_deferred(blomgpu_ctx *c) {
  AleState *a = (AleState *)c->ale;
  if (!a) return 0;
  int rc;
  if ((rc = h3m_sequence_poll(a->grid, nullptr)) || (rc = h3m_sequence_poll(a->grid_uv, nullptr))) return ale_fail(c, "a column failed", rc);
  return 0;
}
