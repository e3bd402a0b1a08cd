// pgforc -- pressure gradient force, phy/mod_pgforc.F90:438-615, with
// pgforc_geopotential (:95-260).
//
// Kernels (one thread per water column unless noted; planes are read coalesced in i while
// each lane walks its own column in k -- the k-recurrences phi(k) = phi(k+1) - dphi and
// the monotone search indices kup/kum are per-column serial by construction):
//   k_pscan + k_dpudpv      p, dpu, dpv, pu, pv                                 (:450-485)
//   k_pgf_copy_old          *_o copies of the previous PGF fields                (:488-522)
//   k_pgf_phi               bottom-up geopotential and phip at p-points          (:112-134)
//   k_pgf_uv                per u-/v-column: downward index search, delphi at the
//                           mid-layer pressure, pgfx/pgfy and the vertical sums
//                           pgfxm, xixp, xixm; then the normalisation by pbu_p, the
//                           barotropic part removal and the /pb_p scaling        (:140-257, :543-589)
//   sealv                                                                       (:591-595)
// Algorithmic bytes: 15 F.  Roofline: HBM (EOS is ~100 flop per 8-byte load).
#include "blomgpu_internal.h"
#include "eos.h"

int launch_p_dpu_dpv(blomgpu_ctx *c, int off, int with_pupv);

#define EPSILP 1.e-12
#define GRAV 9.806


__global__ void k_pgf_copy_old2d(const DevView *__restrict__ Vp, int n) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < -1 || j > V.jj + 2 || i < 0 || i > V.ii + 1) return;
  const size_t c = t, on = (size_t)(n - 1) * V.nplane;
  if (V.m[I_iu][c]) {
    V.f[F_xixp_o][c] = V.f[F_xixp][c + on];
    V.f[F_xixm_o][c] = V.f[F_xixm][c + on];
    V.f[F_pgfxm_o][c] = V.f[F_pgfxm][c + on];
  }
  if (V.m[I_iv][c]) {
    V.f[F_xiyp_o][c] = V.f[F_xiyp][c + on];
    V.f[F_xiym_o][c] = V.f[F_xiym][c + on];
    V.f[F_pgfym_o][c] = V.f[F_pgfym][c + on];
  }
}

__global__ void k_pgf_copy_old3d(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const int k = blockIdx.y;
  const size_t c = t, ok = (size_t)k * V.nplane, okn = (size_t)(k + nn) * V.nplane;
  if (V.m[I_iu][c]) V.f[F_pgfx_o][c + ok] = V.f[F_pgfx][c + okn];
  if (V.m[I_iv][c]) V.f[F_pgfy_o][c + ok] = V.f[F_pgfy][c + okn];
}

// phi, phip at p-points, j,i = 0..jj/ii, phy/mod_pgforc.F90:112-134.  phip -> wkp0.
__global__ __launch_bounds__(64) void k_pgf_phi(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 0 || j > V.jj || i < 0 || i > V.ii || !V.m[I_ip][t]) return;
  const size_t c = t, np = V.nplane;
  const int kk = V.kk;
  gd_t phi = V.f[F_phi], phip = V.f[F_wkp0];
  gcd_t p = V.f[F_p];
  double ph = phi[c + (size_t)kk * np], php = 0.;
  phip[c + (size_t)kk * np] = 0.;
  double plo = p[c + (size_t)kk * np];
  gcd_t dpn = V.f[F_dp] + (size_t)nn * np + c, tn = V.f[F_temp] + (size_t)nn * np + c, sn = V.f[F_saln] + (size_t)nn * np + c;
  for (int k0 = kk - 1; k0 >= 0; k0 -= COLUMN_U) {            // COLUMN_U levels' loads in flight (blomgpu_internal.h)
    double a[COLUMN_U], b[COLUMN_U], d[COLUMN_U], e[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const size_t o = (size_t)(k0 - u >= 0 ? k0 - u : 0) * np;
      a[u] = p[c + o]; b[u] = dpn[o]; d[u] = tn[o]; e[u] = sn[o];
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 - u;
      if (k >= 0) {
        const double pup = a[u];
        if (!(b[u] < EPSILP)) {
          double dphi, alpu, alpl;
          eos::delphi(pup, plo, d[u], e[u], dphi, alpu, alpl);
          ph = ph - dphi;
          php = php + plo * alpl - pup * alpu;
        }
        phi[c + (size_t)k * np] = ph;
        phip[c + (size_t)k * np] = php;
        plo = pup;
      }
    }
  }
}

// One u- or v-column per thread (blockIdx.y = 0: u, 1: v), phy/mod_pgforc.F90:140-257 and the
// per-column part of :543-589.  `sh` is the plane offset of the "minus" neighbour (-1 or -ni).
// A workgroup is two wavefronts over the same 64 points -- the u-columns and the v-columns -- and the workgroups are
// numbered so that an XCD walks a contiguous eighth of the plane (xcd_block): the p-column records (p, T, S, phi, phi')
// that the u- and the v-column of a point and their row neighbours share are then fetched into one L2 instead of up to
// four (the u- and v-columns used to be separate workgroups: 1.44 GB fetched for ~0.5 GB of distinct bytes).
// COPY: the *_o copy of the previous pressure gradient (:488-522, k_pgf_copy_old3d) rides along -- the kernel overwrites
// pgfx/pgfy(kn) level by level and takes the old value on the way (one launch and one sweep over the field less)
#ifdef BLOM_HOSTEMU
#define PGF_WAVE_ALL(p) false
#else
#define PGF_WAVE_ALL(p) (__all(p) != 0)
#endif
template <bool PAIR, bool COPY>
__global__ __launch_bounds__(128) void k_pgf_uv(const DevView *__restrict__ Vp, int n, int nn, int reuse KPROF_ARGS) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t = PAIR ? bx_ * 64 + (threadIdx.x & 63) : blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = PAIR ? threadIdx.x >= 64 : blockIdx.y == 1;
  const size_t c = t, np = V.nplane;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t mns = isv ? c - V.ni : c - 1;
  const int kk = V.kk;
  gcd_t p = V.f[F_p], phi = V.f[F_phi], phip = V.f[F_wkp0];
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
  gcd_t pz = isv ? V.f[F_pv] : V.f[F_pu];
  gcd_t dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gd_t pgf = (isv ? V.f[F_pgfy] : V.f[F_pgfx]) + (size_t)nn * np;
  gd_t pgf_o = isv ? V.f[F_pgfy_o] : V.f[F_pgfx_o];
  // The layers kp, km of the two scalar columns that hold the pressure of the velocity point's layer centre move
  // upwards with k, usually by one.  What the level needs of layer kp -- p above and below, T, S, phi, phi' -- is kept
  // in registers together with the same record of layer kp-1, re-loaded in the background when kp moves; the fixed-index
  // loads of level k-1 are issued while level k is worked on.  Without this every level waits for three to four
  // dependent memory round trips (0.34 ms per call for 3500 wavefronts).
  struct Rec { double pu, pl, t, s, ph, php; };
  auto load_rec = [&](size_t col, int kq) {
    const int kc = kq < 1 ? 1 : kq;
    Rec r;
    r.pu = p[col + (size_t)(kc - 1) * np]; r.pl = p[col + (size_t)kc * np];
    r.t = temp[col + (size_t)(kc - 1) * np]; r.s = saln[col + (size_t)(kc - 1) * np];
    r.ph = phi[col + (size_t)kc * np]; r.php = phip[col + (size_t)kc * np];
    return r;
  };
  [[maybe_unused]] const int wid = PAIR ? (int)bx_ * 2 + (int)(threadIdx.x >> 6) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
  KPROF_MARK(wid, 0);
  int kp = kk, km = kk;                  // kup/kum (1-based layer indices as in the reference)
  Rec rp = load_rec(c, kp), rp1 = load_rec(c, kp - 1), rm = load_rec(mns, km), rm1 = load_rec(mns, km - 1);
  double xip = 0., xim = 0., pgfm = 0.;
  double dpk_n = dpz[c + (size_t)(kk - 1) * np], pz_n = pz[c + (size_t)kk * np];
  double pck = p[c + (size_t)kk * np], pmk = p[mns + (size_t)kk * np];
  double pck1_n = p[c + (size_t)(kk - 1) * np], pmk1_n = p[mns + (size_t)(kk - 1) * np];
  double old_n = COPY ? pgf[c + (size_t)(kk - 1) * np] : 0.;
  double prs_prev = -1., dphip = 0., alpup = 0., alplp = 0., dphim = 0., alpum = 0., alplm = 0.;
  for (int k = kk; k >= 1; k--) {
    const double dpk = dpk_n, pzk = pz_n, pck1 = pck1_n, pmk1 = pmk1_n, old = old_n;
    if (k > 1) {                                             // level k-1's fixed-index loads
      dpk_n = dpz[c + (size_t)(k - 2) * np]; pz_n = pz[c + (size_t)(k - 1) * np];
      pck1_n = p[c + (size_t)(k - 2) * np]; pmk1_n = p[mns + (size_t)(k - 2) * np];
      if (COPY) old_n = pgf[c + (size_t)(k - 2) * np];
    }
    if (COPY) pgf_o[c + (size_t)(k - 1) * np] = old;
    const double prs = pzk - .5 * dpk;
#ifdef BLOM_KPROF
    const int kp0_ = kp, km0_ = km;
#endif
    while (rp.pu > prs) { kp--; rp = rp1; rp1 = load_rec(c, kp - 1); }
    while (rm.pu > prs) { km--; rm = rm1; rm1 = load_rec(mns, km - 1); }
#ifdef BLOM_KPROF
    KPROF_ADD(wid, 4, (kp0_ - kp) + (km0_ - km));
    if (kp0_ - kp > 1 || km0_ - km > 1) KPROF_ADD(wid, 5, 1);
    if (k == kk) KPROF_MARK(wid, 1);
#endif
    const double pplo = rp.pl, pmlo = rm.pl;
    // a massless velocity layer repeats the previous level's mid-layer pressure and with it both records: the equation-of-state values
    // of the previous level are this level's, bit for bit (see k_pgf_uv_ring, REUSE)
    if (!(reuse && PGF_WAVE_ALL(prs == prs_prev))) {
      eos::delphi(prs, pplo, rp.t, rp.s, dphip, alpup, alplp);
      eos::delphi(prs, pmlo, rm.t, rm.s, dphim, alpum, alplm);
    }
    prs_prev = prs;
    double cp = .25 * (pck + pck1);
    double cm = .25 * (pmk + pmk1);
    const double q = prs / (cp + cm);
    cp = q * cp;
    cm = q * cm;
    const double phi_p = rp.ph - dphip;
    xip = xip + (rp.php + pplo * alplp - cp * (alpup - alpum)) * dpk;
    const double phi_m = rm.ph - dphim;
    xim = xim + (rm.php + pmlo * alplm - cm * (alpum - alpup)) * dpk;
    const double g = -(phi_p - phi_m);
    pgf[c + (size_t)(k - 1) * np] = g;
    pgfm = pgfm + g * dpk;
    pck = pck1; pmk = pmk1;
  }
  KPROF_MARK(wid, 2);
  // :543-589
  const double q = 1. / (isv ? V.f[F_pbv_p][c] : V.f[F_pbu_p][c]);
  pgfm = pgfm * q;
  xip = xip * q;
  xim = xim * q;
  for (int k0 = 0; k0 < kk; k0 += COLUMN_U) {
    double a0[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) a0[u] = pgf[c + (size_t)(k0 + u < kk ? k0 + u : kk - 1) * np];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++)
      if (k0 + u < kk) pgf[c + (size_t)(k0 + u) * np] = a0[u] - pgfm;
  }
  const size_t on = (size_t)(n - 1) * np;
  (isv ? V.f[F_pgfym] : V.f[F_pgfxm])[c + on] = pgfm + xip - xim;
  (isv ? V.f[F_xiyp] : V.f[F_xixp])[c + on] = xip / V.f[F_pb_p][c];
  (isv ? V.f[F_xiym] : V.f[F_xixm])[c + on] = xim / V.f[F_pb_p][mns];
  KPROF_MARK(wid, 3);
}

// k_pgf_uv with every load of the level loop statically countable (round 6).  In k_pgf_uv above the record of the layer above
// (rp1 / rm1) is re-loaded inside the data-dependent `while`: the lanes of a wave take that path at different levels, a wave has ONE
// load counter, and the compiler cannot count loads through a divergent loop -- so it waits for vmcnt(0) at every level, i.e. for the
// YOUNGEST load of the wave, the fixed-index prefetch of the next level included: one full memory round trip per level.  Here a
// level issues, unconditionally and in straight-line code, (a) the fixed-index loads of level k-1 and (b) the record of layer kp-2 /
// km-2 of the two scalar columns (the one a move by one layer will need NEXT level), into the register set the previous level does
// not use (the loop is unrolled by two: no copies of registers whose loads are in flight).  A move by one layer is two selects:
// r0 = r1 now, r1 = that speculative record at the start of the next level, when it has had a level's time to arrive -- the wait
// is vmcnt(n > 0), for the loads of the PREVIOUS level.  A move by two or more layers in one level (massless layers of the scalar
// column, the first level over a shallow bottom) takes the slow path: the layer is found with four p-loads in flight per round and
// its two records are loaded and waited for.  Arithmetic, its order and the selected layers are those of k_pgf_uv: bit-identical.
struct PgfRec { double pu, t, s, ph, php; };      // layer q of a scalar column: p(q), T(q), S(q), phi(q+1), phi'(q+1); p(q+1) is the layer below's pu
struct PgfFix { double dpk, pzk, pck1, pmk1, old; };
#define WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)     // s_waitcnt vmcnt(0) (gfx9 encoding: expcnt, lgkmcnt at their maxima)

// DB: the speculative records and the fixed-index loads double-buffered over two unrolled levels (the level's loads are issued BEFORE
// the wait for the previous level's: period (latency + arithmetic) / 2 per level, ~50 more VGPRs); !DB: one set (merge, then issue)
// REUSE: where a level's mid-layer pressure equals the previous level's (a massless velocity layer: dp = 0 -- two thirds of all levels
// of the channel's bench state, and 28 of a wavefront's 53 levels for all of its lanes at once) the two scalar-column records are
// the previous level's too, so both equation-of-state evaluations would reproduce the previous level's values bit for bit: a
// wave-uniform branch skips them (the sums still receive their -- zero-thickness -- terms, in the reference's order).
template <bool PAIR, bool COPY, bool DB, bool REUSE>
__device__ __forceinline__ void pgf_uv_ring_body(const DevView *__restrict__ Vp, int n, int nn KPROF_ARGS) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t = PAIR ? bx_ * 64 + (threadIdx.x & 63) : blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = PAIR ? threadIdx.x >= 64 : blockIdx.y == 1;
  const size_t c = t, np = V.nplane;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t mns = isv ? c - V.ni : c - 1;
  const int kk = V.kk;
  gcd_t p = V.f[F_p], phi = V.f[F_phi], phip = V.f[F_wkp0];
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
  gcd_t pz = isv ? V.f[F_pv] : V.f[F_pu];
  gcd_t dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gd_t pgf = (isv ? V.f[F_pgfy] : V.f[F_pgfx]) + (size_t)nn * np;
  [[maybe_unused]] gd_t pgf_o = isv ? V.f[F_pgfy_o] : V.f[F_pgfx_o];
  auto load_rec = [&](size_t col, int kq) {
    const int kc = kq < 1 ? 1 : kq;
    PgfRec r;
    r.pu = p[col + (size_t)(kc - 1) * np];
    r.t = temp[col + (size_t)(kc - 1) * np]; r.s = saln[col + (size_t)(kc - 1) * np];
    r.ph = phi[col + (size_t)kc * np]; r.php = phip[col + (size_t)kc * np];
    return r;
  };
  auto load_fix = [&](int k) {                     // what level k needs at fixed indices
    const int kc = k < 1 ? 1 : k;
    PgfFix f;
    f.dpk = dpz[c + (size_t)(kc - 1) * np]; f.pzk = pz[c + (size_t)kc * np];
    f.pck1 = p[c + (size_t)(kc - 1) * np]; f.pmk1 = p[mns + (size_t)(kc - 1) * np];
    f.old = COPY ? pgf[c + (size_t)(kc - 1) * np] : 0.;
    return f;
  };
  auto sel = [](bool a, const PgfRec &x, const PgfRec &y) {
    PgfRec r;
    r.pu = a ? x.pu : y.pu; r.t = a ? x.t : y.t; r.s = a ? x.s : y.s; r.ph = a ? x.ph : y.ph; r.php = a ? x.php : y.php;
    return r;
  };
  // the layer of column `col` that holds prs, searched upwards from layer q0 (p(q0+1) > prs is known): the largest q <= q0 with
  // p(q) <= prs, four interface pressures in flight per round
  auto find = [&](size_t col, int q0, double prs) {
    for (;;) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int q = q0 - u < 1 ? 1 : q0 - u; v[u] = p[col + (size_t)(q - 1) * np]; }
      int nup = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) nup += v[u] > prs ? 1 : 0;
      if (nup < 4 || q0 - 4 < 1) { const int q = q0 - nup; return q < 1 ? 1 : q; }
      q0 -= 4;
    }
  };
  [[maybe_unused]] const int wid = PAIR ? (int)bx_ * 2 + (int)(threadIdx.x >> 6) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
  KPROF_MARK(wid, 0);
  int kp = kk, km = kk;                  // kup/kum (1-based layer indices as in the reference)
  PgfRec rp = load_rec(c, kp), rp1 = load_rec(c, kp - 1), rm = load_rec(mns, km), rm1 = load_rec(mns, km - 1);
  double pplo = p[c + (size_t)kk * np], pmlo = p[mns + (size_t)kk * np];      // p(kp+1), p(km+1)
  PgfRec spA = rp1, smA = rm1, spB = rp1, smB = rm1;
  PgfFix fxA = load_fix(kk), fxB = fxA;
  bool advp = false, advm = false;
  double xip = 0., xim = 0., pgfm = 0.;
  double pck = pplo, pmk = pmlo;
  double prs_prev = -1., dphip = 0., alpup = 0., alplp = 0., dphim = 0., alpum = 0., alplm = 0.;
  auto level = [&](const int k, PgfFix &fx, PgfFix &fxn, PgfRec &spo, PgfRec &smo, PgfRec &spn, PgfRec &smn) {
    if (DB) {                            // this level's loads, all unconditional, then the previous level's records take their place
      fxn = load_fix(k - 1);
      spn = load_rec(c, kp - 2);
      smn = load_rec(mns, km - 2);
      rp1 = sel(advp, spo, rp1);
      rm1 = sel(advm, smo, rm1);
    } else {
      rp1 = sel(advp, spo, rp1);
      rm1 = sel(advm, smo, rm1);
      fx = fxn;
      fxn = load_fix(k - 1);
      spo = load_rec(c, kp - 2);
      smo = load_rec(mns, km - 2);
    }
    if (COPY) pgf_o[c + (size_t)(k - 1) * np] = fx.old;
    const double dpk = fx.dpk, pck1 = fx.pck1, pmk1 = fx.pmk1;
    const double prs = fx.pzk - .5 * dpk;
    const bool a1p = rp.pu > prs, a1m = rm.pu > prs;
    const bool a2p = a1p && rp1.pu > prs, a2m = a1m && rm1.pu > prs;
    advp = a1p; advm = a1m;
    if (a1p) { kp--; pplo = rp.pu; rp = rp1; }
    if (a1m) { km--; pmlo = rm.pu; rm = rm1; }
    if (a2p || a2m) {                    // two or more layers at once: the slow path ends with nothing in flight
      KPROF_ADD(wid, 5, 1);
      if (a2p) { kp = find(c, kp - 1, prs); rp = load_rec(c, kp); rp1 = load_rec(c, kp - 1); pplo = p[c + (size_t)kp * np]; advp = false; }
      if (a2m) { km = find(mns, km - 1, prs); rm = load_rec(mns, km); rm1 = load_rec(mns, km - 1); pmlo = p[mns + (size_t)km * np]; advm = false; }
      WAIT_VM0();
    }
    if (k == kk) KPROF_MARK(wid, 1);
    KPROF_ADD(wid, 4, (a1p ? 1 : 0) + (a1m ? 1 : 0));
    if (!(REUSE && PGF_WAVE_ALL(prs == prs_prev))) {
      eos::delphi(prs, pplo, rp.t, rp.s, dphip, alpup, alplp);
      eos::delphi(prs, pmlo, rm.t, rm.s, dphim, alpum, alplm);
    }
    prs_prev = prs;
    double cp = .25 * (pck + pck1);
    double cm = .25 * (pmk + pmk1);
    const double q = prs / (cp + cm);
    cp = q * cp;
    cm = q * cm;
    const double phi_p = rp.ph - dphip;
    xip = xip + (rp.php + pplo * alplp - cp * (alpup - alpum)) * dpk;
    const double phi_m = rm.ph - dphim;
    xim = xim + (rm.php + pmlo * alplm - cm * (alpum - alpup)) * dpk;
    const double g = -(phi_p - phi_m);
    pgf[c + (size_t)(k - 1) * np] = g;
    pgfm = pgfm + g * dpk;
    pck = pck1; pmk = pmk1;
  };
  if (DB) {
    int k = kk;
    for (; k >= 2; k -= 2) {
      level(k, fxA, fxB, spA, smA, spB, smB);
      level(k - 1, fxB, fxA, spB, smB, spA, smA);
    }
    if (k == 1) level(1, fxA, fxB, spA, smA, spB, smB);
  } else {
    for (int k = kk; k >= 1; k--) level(k, fxB, fxA, spA, smA, spA, smA);
  }
  KPROF_MARK(wid, 2);
  // :543-589
  const double q = 1. / (isv ? V.f[F_pbv_p][c] : V.f[F_pbu_p][c]);
  pgfm = pgfm * q;
  xip = xip * q;
  xim = xim * q;
  for (int k0 = 0; k0 < kk; k0 += COLUMN_U) {
    double a0[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) a0[u] = pgf[c + (size_t)(k0 + u < kk ? k0 + u : kk - 1) * np];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++)
      if (k0 + u < kk) pgf[c + (size_t)(k0 + u) * np] = a0[u] - pgfm;
  }
  const size_t on = (size_t)(n - 1) * np;
  (isv ? V.f[F_pgfym] : V.f[F_pgfxm])[c + on] = pgfm + xip - xim;
  (isv ? V.f[F_xiyp] : V.f[F_xixp])[c + on] = xip / V.f[F_pb_p][c];
  (isv ? V.f[F_xiym] : V.f[F_xixm])[c + on] = xim / V.f[F_pb_p][mns];
  KPROF_MARK(wid, 3);
}

template <bool PAIR, bool COPY, bool DB, bool REUSE>
__global__ __launch_bounds__(128) void k_pgf_uv_ring(const DevView *__restrict__ Vp, int n, int nn KPROF_ARGS) {
#ifdef BLOM_KPROF
  pgf_uv_ring_body<PAIR, COPY, DB, REUSE>(Vp, n, nn, kprof, kprof_words);
#else
  pgf_uv_ring_body<PAIR, COPY, DB, REUSE>(Vp, n, nn);
#endif
}
// W4: held to 128 VGPRs (four waves per SIMD: all 3 510 wavefronts of the channel resident at once; the compiler spills 26 - 124 registers)
template <bool PAIR, bool COPY, bool DB, bool REUSE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pgf_uv_ring_w4(const DevView *__restrict__ Vp, int n, int nn KPROF_ARGS) {
#ifdef BLOM_KPROF
  pgf_uv_ring_body<PAIR, COPY, DB, REUSE>(Vp, n, nn, kprof, kprof_words);
#else
  pgf_uv_ring_body<PAIR, COPY, DB, REUSE>(Vp, n, nn);
#endif
}

// k_pgf_uv_next: the lean form of the ring (round 6, after the per-wavefront timestamps of tools/kprof_waves.py).  k_pgf_uv_ring's
// wavefronts live half as long as k_pgf_uv's (216 against 403 us), but at 140 - 176 VGPRs only three of them fit a SIMD: 3 072 of the
// channel's 3 442 start at once and the launch waits for a second round.  Here a scalar column keeps ONE full record (its current
// layer kp) and one speculative record, that of layer kp - 1, requested unconditionally every level right after the level's move is
// known -- a level's time before a move can need it.  A move by one layer copies it; by two or more: the slow path (four interface
// pressures in flight per round, then both records).  Still every load of the fast path is in straight-line code (counted waits).
// LAZY: the speculative record is requested again only by the lanes whose column moved at this level (it is still valid for the
// others: at two thirds of the levels -- massless velocity layers -- nothing moves); the loads stay in the level's straight-line
// part, under the lanes' mask.
template <bool PAIR, bool COPY, bool LAZY>
__global__ __launch_bounds__(128) void k_pgf_uv_next(const DevView *__restrict__ Vp, int n, int nn KPROF_ARGS) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t = PAIR ? bx_ * 64 + (threadIdx.x & 63) : blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = PAIR ? threadIdx.x >= 64 : blockIdx.y == 1;
  const size_t c = t, np = V.nplane;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t mns = isv ? c - V.ni : c - 1;
  const int kk = V.kk;
  gcd_t p = V.f[F_p], phi = V.f[F_phi], phip = V.f[F_wkp0];
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np;
  gcd_t pz = isv ? V.f[F_pv] : V.f[F_pu];
  gcd_t dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gd_t pgf = (isv ? V.f[F_pgfy] : V.f[F_pgfx]) + (size_t)nn * np;
  [[maybe_unused]] gd_t pgf_o = isv ? V.f[F_pgfy_o] : V.f[F_pgfx_o];
  auto load_rec = [&](size_t col, int kq) {
    const int kc = kq < 1 ? 1 : kq;
    PgfRec r;
    r.pu = p[col + (size_t)(kc - 1) * np];
    r.t = temp[col + (size_t)(kc - 1) * np]; r.s = saln[col + (size_t)(kc - 1) * np];
    r.ph = phi[col + (size_t)kc * np]; r.php = phip[col + (size_t)kc * np];
    return r;
  };
  auto load_fix = [&](int k) {
    const int kc = k < 1 ? 1 : k;
    PgfFix f;
    f.dpk = dpz[c + (size_t)(kc - 1) * np]; f.pzk = pz[c + (size_t)kc * np];
    f.pck1 = p[c + (size_t)(kc - 1) * np]; f.pmk1 = p[mns + (size_t)(kc - 1) * np];
    f.old = COPY ? pgf[c + (size_t)(kc - 1) * np] : 0.;
    return f;
  };
  auto find = [&](size_t col, int q0, double prs) {      // the largest q <= q0 with p(q) <= prs
    for (;;) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int q = q0 - u < 1 ? 1 : q0 - u; v[u] = p[col + (size_t)(q - 1) * np]; }
      int nup = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) nup += v[u] > prs ? 1 : 0;
      if (nup < 4 || q0 - 4 < 1) { const int q = q0 - nup; return q < 1 ? 1 : q; }
      q0 -= 4;
    }
  };
  [[maybe_unused]] const int wid = PAIR ? (int)bx_ * 2 + (int)(threadIdx.x >> 6) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
  KPROF_MARK(wid, 0);
  int kp = kk, km = kk;                  // kup/kum (1-based layer indices as in the reference)
  PgfRec rp = load_rec(c, kp), sp = load_rec(c, kp - 1), rm = load_rec(mns, km), sm = load_rec(mns, km - 1);
  double pplo = p[c + (size_t)kk * np], pmlo = p[mns + (size_t)kk * np];      // p(kp+1), p(km+1)
  PgfFix fxn = load_fix(kk);
  double xip = 0., xim = 0., pgfm = 0.;
  double pck = pplo, pmk = pmlo, g_prev = 0.;
  for (int k = kk; k >= 1; k--) {
    // the one wait of a level: for the loads the previous level issued before its arithmetic.  Explicit, so that the compiler's own
    // bookkeeping starts every level with nothing in flight (with the lanes' masks on the loads it would otherwise place
    // conservative waits between the loads and the arithmetic)
    WAIT_VM0();
    const PgfFix fx = fxn;
    const double dpk = fx.dpk, pck1 = fx.pck1, pmk1 = fx.pmk1;
    const double prs = fx.pzk - .5 * dpk;
    const bool a1p = rp.pu > prs, a1m = rm.pu > prs;
    const bool a2p = a1p && sp.pu > prs, a2m = a1m && sm.pu > prs;
    if (a1p) { kp--; pplo = rp.pu; rp = sp; }
    if (a1m) { km--; pmlo = rm.pu; rm = sm; }
    if (a2p || a2m) {                    // two or more layers at once
      KPROF_ADD(wid, 5, 1);
      if (a2p) { kp = find(c, kp - 1, prs); rp = load_rec(c, kp); pplo = p[c + (size_t)kp * np]; }
      if (a2m) { km = find(mns, km - 1, prs); rm = load_rec(mns, km); pmlo = p[mns + (size_t)km * np]; }
      WAIT_VM0();
    }
    // the level's stores go out BEFORE its loads -- the previous level's result, kept in a register until here, and the *_o copy --
    // so that the youngest operations in flight at the next level's top are loads it needs anyway (a store issued at the end of the
    // level would be the wave's youngest: waiting for the records would wait for its acknowledgement, a write round trip per level)
    if (k < kk) pgf[c + (size_t)k * np] = g_prev;
    if (COPY) pgf_o[c + (size_t)(k - 1) * np] = fx.old;
    // this level's loads: the fixed-index ones of level k - 1 and the records a move at level k - 1 would need
    fxn = load_fix(k - 1);
    if (!LAZY || a1p) sp = load_rec(c, kp - 1);
    if (!LAZY || a1m) sm = load_rec(mns, km - 1);
    // (the scheduler minimises register pressure: without the barrier it moves these loads below the equation of state, next to
    // their first use -- the opposite of what they are issued here for)
    __builtin_amdgcn_sched_barrier(0);
    if (k == kk) KPROF_MARK(wid, 1);
    KPROF_ADD(wid, 4, (a1p ? 1 : 0) + (a1m ? 1 : 0));
    double dphip, alpup, alplp, dphim, alpum, alplm;
    eos::delphi(prs, pplo, rp.t, rp.s, dphip, alpup, alplp);
    eos::delphi(prs, pmlo, rm.t, rm.s, dphim, alpum, alplm);
    double cp = .25 * (pck + pck1);
    double cm = .25 * (pmk + pmk1);
    const double q = prs / (cp + cm);
    cp = q * cp;
    cm = q * cm;
    const double phi_p = rp.ph - dphip;
    xip = xip + (rp.php + pplo * alplp - cp * (alpup - alpum)) * dpk;
    const double phi_m = rm.ph - dphim;
    xim = xim + (rm.php + pmlo * alplm - cm * (alpum - alpup)) * dpk;
    const double g = -(phi_p - phi_m);
    g_prev = g;
    pgfm = pgfm + g * dpk;
    pck = pck1; pmk = pmk1;
  }
  pgf[c] = g_prev;
  KPROF_MARK(wid, 2);
  // :543-589
  const double q = 1. / (isv ? V.f[F_pbv_p][c] : V.f[F_pbu_p][c]);
  pgfm = pgfm * q;
  xip = xip * q;
  xim = xim * q;
  for (int k0 = 0; k0 < kk; k0 += COLUMN_U) {
    double a0[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) a0[u] = pgf[c + (size_t)(k0 + u < kk ? k0 + u : kk - 1) * np];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++)
      if (k0 + u < kk) pgf[c + (size_t)(k0 + u) * np] = a0[u] - pgfm;
  }
  const size_t on = (size_t)(n - 1) * np;
  (isv ? V.f[F_pgfym] : V.f[F_pgfxm])[c + on] = pgfm + xip - xim;
  (isv ? V.f[F_xiyp] : V.f[F_xixp])[c + on] = xip / V.f[F_pb_p][c];
  (isv ? V.f[F_xiym] : V.f[F_xixm])[c + on] = xim / V.f[F_pb_p][mns];
  KPROF_MARK(wid, 3);
}

// ---- pgforc_dynamic_enthalpy, phy/mod_pgforc.F90:269-412 ---------------------------------------------
// work-space slots of the layer potentials
enum { DH_POT = 0, DH_POTPB, DH_A, DH_T, DH_ALPR, DH_NSLOT };
#define P0_DYNH 0.           // phy/mod_pgforc.F90:49
#define ONEMM_ 9.806

// per p-column, j,i = 0..jj/ii: bottom-up potentials (:282-312) and the derivatives (:318-341)
__global__ void k_pgf_dynh_col(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 0 || j > V.jj || i < 0 || i > V.ii || !V.m[I_ip][t]) return;
  const size_t c = t, np = V.nplane;
  const int kk = V.kk;
  const double pref = V.P.pref;
  gd_t phi = V.f[F_phi];
  gcd_t p = V.f[F_p];
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, saln = V.f[F_saln] + (size_t)nn * np, dp = V.f[F_dp] + (size_t)nn * np;
#define LV(a, k) (a)[c + (size_t)((k)-1) * np]
  double pot, potpb, ph = LV(phi, kk + 1);
  {
    const double tk = LV(temp, kk), sk = LV(saln, kk), pl = LV(p, kk + 1);
    pot = ph + eos::p_alpha(P0_DYNH, pl, tk, sk);
    potpb = eos::alp(pl, tk, sk) * pl;
    ph = ph + eos::p_alpha(LV(p, kk), pl, tk, sk);
    LV(WK(V, DH_POT), kk) = pot;
    LV(WK(V, DH_POTPB), kk) = potpb;
    LV(phi, kk) = ph;
  }
  for (int k = kk - 1; k >= 1; k--) {
    const double tk = LV(temp, k), sk = LV(saln, k), tl = LV(temp, k + 1), sl = LV(saln, k + 1), pl = LV(p, k + 1);
    pot = pot + eos::p_alpha(P0_DYNH, pl, tk, sk) - eos::p_alpha(P0_DYNH, pl, tl, sl);
    potpb = potpb + (eos::alp(pl, tk, sk) - eos::alp(pl, tl, sl)) * pl;
    ph = ph + eos::p_alpha(LV(p, k), pl, tk, sk);
    LV(WK(V, DH_POT), k) = pot;
    LV(WK(V, DH_POTPB), k) = potpb;
    LV(phi, k) = ph;
  }
  for (int k = 1; k <= kk; k++) {
    const double tk = LV(temp, k), sk = LV(saln, k);
    double da = 0., dt = 0.;
    if (!(LV(dp, k) < ONEMM_)) {
      double ts_t, ts_s;
      eos::dynh_derivatives(P0_DYNH, LV(p, k), LV(p, k + 1), tk, sk, ts_t, ts_s);
      da = ts_s / eos::dalpds(pref, tk, sk);
      dt = ts_t - da * eos::dalpdt(pref, tk, sk);
    }
    LV(WK(V, DH_A), k) = da;
    LV(WK(V, DH_T), k) = dt;
    LV(WK(V, DH_ALPR), k) = eos::alp(pref, tk, sk);
  }
#undef LV
}

// per u-/v-column (blockIdx.y = 0: u, 1: v): layer PGF and the vertical sums, :345-410, then :543-589
__global__ void k_pgf_dynh_uv(const DevView *__restrict__ Vp, int n, int nn) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = blockIdx.y == 1;
  const size_t c = t, np = V.nplane;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t mns = isv ? c - V.ni : c - 1;
  const int kk = V.kk;
  gcd_t temp = V.f[F_temp] + (size_t)nn * np, dp = V.f[F_dp] + (size_t)nn * np;
  gcd_t dpz = (isv ? V.f[F_dpv] : V.f[F_dpu]) + (size_t)nn * np;
  gd_t pgf = (isv ? V.f[F_pgfy] : V.f[F_pgfx]) + (size_t)nn * np;
  [[maybe_unused]] gd_t pgf_o = isv ? V.f[F_pgfy_o] : V.f[F_pgfx_o];
  gcd_t pot = WK(V, DH_POT), potpb = WK(V, DH_POTPB), da = WK(V, DH_A), dt = WK(V, DH_T), ar = WK(V, DH_ALPR);
  double xip = 0., xim = 0., pgfm = 0.;
  for (int k = kk; k >= 1; k--) {
    const size_t o = (size_t)(k - 1) * np;
    double g = -(pot[c + o] - pot[mns + o]);
    if (dp[mns + o] >= ONEMM_ && dp[c + o] >= ONEMM_)
      g = g + .5 * ((dt[mns + o] + dt[c + o]) * (temp[c + o] - temp[mns + o]) + (da[mns + o] + da[c + o]) * (ar[c + o] - ar[mns + o]));
    pgf[c + o] = g;
    const double dpk = dpz[c + o];
    pgfm = pgfm + g * dpk;
    xim = xim + potpb[mns + o] * dpk;
    xip = xip + potpb[c + o] * dpk;
  }
  const double q = 1. / (isv ? V.f[F_pbv_p][c] : V.f[F_pbu_p][c]);
  pgfm = pgfm * q;
  xip = xip * q;
  xim = xim * q;
  for (int k = 0; k < kk; k++) pgf[c + (size_t)k * np] = pgf[c + (size_t)k * np] - pgfm;
  const size_t on = (size_t)(n - 1) * np;
  (isv ? V.f[F_pgfym] : V.f[F_pgfxm])[c + on] = pgfm + xip - xim;
  (isv ? V.f[F_xiyp] : V.f[F_xixp])[c + on] = xip / V.f[F_pb_p][c];
  (isv ? V.f[F_xiym] : V.f[F_xixm])[c + on] = xim / V.f[F_pb_p][mns];
}

__global__ void k_pgf_sealv(const DevView *__restrict__ Vp) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int i = t % V.ni - (NBDY - 1), j = t / V.ni - (NBDY - 1);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][t]) return;
  V.f[F_sealv][t] = V.f[F_phi][t] / GRAV;
}

int st_pgforc(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (int rc = launch_p_dpu_dpv(c, nn, 1)) return rc;
  hipLaunchKernelGGL(k_pgf_copy_old2d, plane_grid(h), dim3(256), 0, c->stream, c->d, n);
  const bool copy_fused = h.P.pgfmth == 0 && c->pgf_copy_fused && !c->pgf_uv_pair;
  if (!copy_fused) hipLaunchKernelGGL(k_pgf_copy_old3d, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
  {
    TimeScope ts(c, "pgforc");
    if (h.P.pgfmth == 0) hipLaunchKernelGGL(k_pgf_phi, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn);
    else hipLaunchKernelGGL(k_pgf_dynh_col, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, nn);
    // xctilr(pb_p,1,1,1,1) at :540 precedes the /pb_p(i-1,j) scaling done inside k_pgf_uv
    if (int rc = st_xctilr(c, h.f[F_pb_p], 1, 1, 1, 1, 1)) return rc;
    if (h.P.pgfmth == 0) {
      TimeScope tk(c, "k_pgf_uv");
      if (c->pgf_uv_ring >= 5 && c->pgf_uv_ring <= 8 && copy_fused) {
        if (c->pgf_uv_ring == 5) hipLaunchKernelGGL((k_pgf_uv_next<false, true, false>), plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, n, nn KPROF_PASS(1));
        else if (c->pgf_uv_ring == 6) hipLaunchKernelGGL((k_pgf_uv_next<true, true, false>), plane_grid(h, 1, 64), dim3(128), 0, c->stream, c->d, n, nn KPROF_PASS(1));
        else if (c->pgf_uv_ring == 7) hipLaunchKernelGGL((k_pgf_uv_next<false, true, true>), plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, n, nn KPROF_PASS(1));
        else hipLaunchKernelGGL((k_pgf_uv_next<true, true, true>), plane_grid(h, 1, 64), dim3(128), 0, c->stream, c->d, n, nn KPROF_PASS(1));
      }
      else if (c->pgf_uv_ring && copy_fused) {
        // pgf_uv_ring: 1 = separate u / v workgroups, 2 = paired + XCD-contiguous; + 2 = the double-buffered form
        const dim3 gs = plane_grid(h, 2, 64), gp = plane_grid(h, 1, 64);
        // + 10: with the reuse of the previous level's equation-of-state values (REUSE); + 100: held to four waves per SIMD (W4)
        const int var = c->pgf_uv_ring % 10, reuse = (c->pgf_uv_ring / 10) % 10, w4 = c->pgf_uv_ring / 100;
#define PGF_LAUNCH(P, D, R)                                                                                                   \
  do {                                                                                                                        \
    if (w4) hipLaunchKernelGGL((k_pgf_uv_ring_w4<P, true, D, R>), P ? gp : gs, dim3(P ? 128 : 64), 0, c->stream, c->d, n, nn KPROF_PASS(1));  \
    else hipLaunchKernelGGL((k_pgf_uv_ring<P, true, D, R>), P ? gp : gs, dim3(P ? 128 : 64), 0, c->stream, c->d, n, nn KPROF_PASS(1));       \
  } while (0)
        if (reuse) {
          if (var == 1) PGF_LAUNCH(false, false, true); else if (var == 2) PGF_LAUNCH(true, false, true);
          else if (var == 3) PGF_LAUNCH(false, true, true); else PGF_LAUNCH(true, true, true);
        } else {
          if (var == 1) PGF_LAUNCH(false, false, false); else if (var == 2) PGF_LAUNCH(true, false, false);
          else if (var == 3) PGF_LAUNCH(false, true, false); else PGF_LAUNCH(true, true, false);
        }
#undef PGF_LAUNCH
      }
      else if (c->pgf_uv_pair) hipLaunchKernelGGL((k_pgf_uv<true, false>), plane_grid(h, 1, 64), dim3(128), 0, c->stream, c->d, n, nn, c->pgf_reuse KPROF_PASS(1));
      else if (copy_fused) hipLaunchKernelGGL((k_pgf_uv<false, true>), plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, n, nn, c->pgf_reuse KPROF_PASS(1));
      else hipLaunchKernelGGL((k_pgf_uv<false, false>), plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, n, nn, c->pgf_reuse KPROF_PASS(1));
    }
    else hipLaunchKernelGGL(k_pgf_dynh_uv, plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, n, nn);
  }
  hipLaunchKernelGGL(k_pgf_sealv, plane_grid(h), dim3(256), 0, c->stream, c->d);
  HIPCHK(c, hipGetLastError());
  return 0;
}
