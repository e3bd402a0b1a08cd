// barotp, fused substep kernel.  One launch advances the barotropic state by one odd+even
// substep pair (or one half of it at the phase boundaries) for a tile of the domain held in LDS.
//
// phy/mod_barotp.F90:387-843: an odd substep is halo update + continuity + u + v, computed 2-3
// cells into the halo; the even substep is continuity + v + u on ranges shrunk accordingly, with
// no communication.  A workgroup therefore loads its TI x TJ tile plus a 3-cell rim of the six
// dynamic planes (pb_t, ubflx_t, vbflx_t at both time levels) into LDS, runs the six sweeps with
// workgroup barriers in between on the shrinking rectangle where inputs are valid (intersected
// with the reference's own loop ranges, so every LDS value equals the reference's array value),
// and writes back the tile interior.  Neighbouring tiles recompute the rim redundantly
// (836/512 points) instead of synchronising grid-wide six times per pair.  Reads come from one
// buffer set and writes go to the other (ping-pong), since a neighbour may still be loading.
// All values are per-point identical to the unfused kernels (same expressions, same order).
#include "blomgpu_internal.h"

// Tile shapes: 32 x 16 (836 points with the rim, 14 wavefronts) is the persistent form's and the default; the smaller ones
// serve contexts whose domain is only a few of those tiles -- the tiles of a multi-GPU decomposition, small grids -- where a
// launch is bound by the sweeps of ONE tile on ONE CU (6.7 of 12.4 us per substep pair) while most CUs are idle.
#define HB 3
// The persistent form's tile is 26 x 16 (704 points with the rim = 11 wavefronts): a workgroup of at most 768 threads may use
// 168 VGPRs, and the kernel needs 158 -- the 32 x 16 tile of rounds 1-2 (896 threads, 128 VGPRs) spilled 80 of them to
// scratch memory, which cost a quarter of every iteration (in-kernel timestamps: the even substep's sweeps took 5.4 us
// against 1.4 us for the odd one's).  208 x 512 points are 8 x 32 = 256 such tiles: one per CU.
// (where the rows do not come out -- 193 = 12 x 16 + 1 leaves a last tile row of one -- 26 x 15; 40 x 16 / 40 x 15, which
// spill 15 registers, for domains of more than 256 such tiles.)
constexpr int bt_threads(int ti, int tj) { return ((ti + 2 * HB) * (tj + 2 * HB) + 63) / 64 * 64; }
struct BtShape { int ti, tj; };

struct PairArgs {
  int m, n, ml, nl;          // baroclinic levels m,n; barotropic levels at the start of the launch
  double wo[2], wm[2], wn[2];  // time weights of the odd [0] and even [1] substep
  int do_odd, do_even;
  int src;                   // 0: read *_t write *_t2, 1: the other way round
  int fold_halo;             // single tile: apply the xctilr rule (wrap / vland) while loading
  long long *prof;           // debug: per-block phase timestamps (nullptr in production)
  // persistent form (k_bt_steps<true>): the launch walks substeps lll0..last itself
  int lll0, last;            // first and last substep of the phase
  double woa, wob, wna, wnb; // time weights wo = woa*l + wob, wn = wna*l + wnb, wm = 1 - wo - wn (:352-360)
  unsigned *flags;           // one word per tile: epoch_base + number of iterations this tile has completed and published
  unsigned epoch_base;       // flags are never reset: every launch counts on from where the previous one stopped
  unsigned *abort_word;      // set by any tile whose wait ran out; every spin also watches it
  // split launches (PERSIST = false, RCCL tiles): 0 all tile columns, 1 only the west-most and east-most
  // column of tiles (they produce the E/W strips the exchange sends), 2 only the columns in between
  int tsel, nbx;
  // arctic patch: also publish the halo cells this tile computed redundantly (the reference's margins,
  // :420-457 etc. run over j = -1..jj+2), so that a following launch needs no halo update -- which with
  // the arctic patch would rewrite the seam row at a point where the reference does not
  int write_margin;
  // RCCL tiles: the E/W rim comes straight from the transport's receive buffers (no unpack launch)
  const double *rim_w, *rim_e;
  int rim_has_w, rim_has_e, rim_per, rim_on;
  // ... and the tiles that hold the outermost HB columns write their part of the next exchange's send
  // strips themselves (no pack launch); N/S rim cells of the columns 1..ii then follow the local rule at load
  double *pack_w, *pack_e;
  int pack_on;
  // k_bt_steps4 walks the substeps of ALL five barotropic phases (:352-977): phase x ends with substep ph_last[x] and has the time weights
  // wo = ph_w[x][0] l + ph_w[x][1], wn = ph_w[x][2] l + ph_w[x][3]; the kernel starts a phase's flux sums at zero (k_bt_zero_sums, :361-379)
  // and ends it with its epilogue (k_bt_epilogue, :847-977) itself
  int ph_last[5] = {0, 0, 0, 0, 0};
  double ph_w[5][4] = {};
};

// PERSIST = false: one odd+even pair (or one half) per launch, neighbours synchronise at the kernel
// boundary.  PERSIST = true: one launch per barotropic phase.  The 45 coefficient planes a launch reads
// are constant over the whole phase, and re-reading them (67 MB per launch on the channel grid, 15 of
// the 20 us of a pair launch) is what the non-persistent form spends its time on; here they stay in
// registers/LDS and only the state moves: after every iteration a tile publishes its interior
// (write-through stores) and a completion count, waits for the counts of its up to 8 neighbours
// (relaxed polls by 8 lanes, one agent-scope acquire, bounded spin with a chip-wide abort word) and
// re-reads just the rim.  Reads alternate between the two buffer sets exactly as the launches did.
// All workgroups must be resident: the launcher checks tiles <= CUs.
template <bool PERSIST, int TI, int TJ>
__global__ void __launch_bounds__(bt_threads(TI, TJ)) k_bt_steps(const DevView *__restrict__ Vp, PairArgs a) {
  constexpr int BI = TI + 2 * HB, BJ = TJ + 2 * HB, NPT = BI * BJ;
  const DevView &V = *Vp;
#ifndef BT_PAD
#define BT_PAD 1
#endif
  __shared__ double s_pb[2][BJ][BI + BT_PAD], s_ub[2][BJ][BI + BT_PAD], s_vb[2][BJ][BI + BT_PAD];
  // coefficients that the momentum equations read at neighbouring points: staged once per launch
  __shared__ double s_pvo[BJ][BI + BT_PAD], s_pvm[BJ][BI + BT_PAD], s_pvn[BJ][BI + BT_PAD], s_sx[BJ][BI + BT_PAD], s_sy[BJ][BI + BT_PAD];
  __shared__ int s_abort;
  const int tid = threadIdx.x;
  const unsigned bx = PERSIST || a.tsel == 0 ? blockIdx.x : (a.tsel == 1 ? (blockIdx.x ? a.nbx - 1 : 0u) : blockIdx.x + 1);
  long long *prof = a.prof ? a.prof + (size_t)(blockIdx.y * gridDim.x + bx) * 16 : nullptr;
  int pslot = 0;
#ifdef BT_PROFILE
#define PROF_MARK() do { if (prof && tid == 0) prof[pslot++] = wall_clock64(); } while (0)
#else
#define PROF_MARK() do { (void)prof; (void)pslot; } while (0)
#endif
  // persistent form, -DBT_PROFILE: [tile][iteration < 16][8] timestamps (100 MHz): 0 top of the iteration, 1 neighbours'
  // counts seen, 2 rim re-read, 3 sweeps done, 4 stores drained, 5 count published
#ifdef BT_PROFILE
#define PMARK(slot) do { if (PERSIST && a.prof && tid == 0 && done_iters < 16u) \
    a.prof[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + done_iters) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PMARK(slot) do { } while (0)
#endif
  PROF_MARK();
  const bool act = tid < NPT;
  const int li = act ? tid % BI : 0, lj = act ? tid / BI : 0;
  // Fortran indices of this thread's point
  const int gi = bx * TI + 1 + li - HB, gj = blockIdx.y * TJ + 1 + lj - HB;
  const int ii = V.ii, jj = V.jj;
  const bool inarr = act && gi >= 1 - NBDY && gi <= ii + NBDY && gj >= 1 - NBDY && gj <= jj + NBDY;
  const size_t np = V.nplane;
  const size_t c = inarr ? (size_t)IDX(V, gi, gj) : 0;
  const gd_t b_pb[2] = {V.f[F_pb_t], V.f[F_pb_t2]}, b_ub[2] = {V.f[F_ubflx_t], V.f[F_ubflx_t2]};
  const gd_t b_vb[2] = {V.f[F_vbflx_t], V.f[F_vbflx_t2]};
  int src = a.src;                       // buffer set that holds the current state
  const size_t om = (size_t)(a.m - 1) * np, on = (size_t)(a.n - 1) * np;
  // source of the state planes: the point itself, or -- single tile, halo point -- what xctilr
  // would have put there (phy/mod_xc.F90:4374-4419): wrapped interior point or vland
  // With the arctic patch (nreg = 2, single tile) the rule is xctilr's for the tripolar seam (k_xctilr_arctic, halo.hip;
  // phy/mod_xc.F90:4277-4371): rows jj.. mirror the rows below the seam about the pole, the source and, for the two
  // vector fields, the sign depend on the grid the field lives on (pb_t: p, ubflx_t: u, vbflx_t: v), and the seam row jj
  // itself is a target on the p- and u-grid and in its second half on the v-grid.
  size_t cs = c, cs_u = c, cs_v = c;
  double sg_u = 1., sg_v = 1.;
  bool land = false;
  if (a.fold_halo && inarr) {
    const bool oi = gi < 1 || gi > ii, oj = gj < 1 || gj > jj;
    if (V.nreg == 2) {
      const int iw = gi < 1 ? gi + ii : (gi > ii ? gi - ii : gi);
      if (gj < 1) land = true;
      else if (gj >= jj) {
        const int d = gj - jj;
        cs = (size_t)IDX(V, ii - (iw - 1) % ii, jj - 1 - d);
        cs_u = (size_t)IDX(V, (ii - (iw - 1)) % ii + 1, jj - 1 - d);
        sg_u = -1.;
        if (d > 0 || iw > ii / 2) { cs_v = (size_t)IDX(V, ii - (iw - 1) % ii, jj - d); sg_v = -1.; }
        else cs_v = (size_t)IDX(V, iw, gj);
      } else if (oi) cs = cs_u = cs_v = (size_t)IDX(V, iw, gj);
    } else if (oi || oj) {
      land = (oi && (V.nreg == 0 || V.nreg == 4)) || (oj && V.nreg <= 2);
      const int is = gi < 1 ? gi + ii : (gi > ii ? gi - ii : gi), js = gj < 1 ? gj + jj : (gj > jj ? gj - jj : gj);
      cs = cs_u = cs_v = (size_t)IDX(V, is, js);
    }
  }
  if (!PERSIST && a.rim_on && !a.fold_halo && inarr && gi >= 1 && gi <= ii && (gj < 1 || gj > jj)) {
    land = V.nreg <= 2;                                   // one tile row: closed in j, or the tile's own periodic wrap
    cs = cs_u = cs_v = (size_t)IDX(V, gi, gj < 1 ? gj + jj : gj - jj);
  }
  const bool ok_src = inarr && !land;
  const bool mine = act && li >= HB && li < HB + TI && lj >= HB && lj < HB + TJ && gi <= ii && gj <= jj;
  // halo cell owned by this tile: the tile that holds the nearest interior point (each halo cell has one owner)
  bool own_halo = false;
  if (a.write_margin && inarr && !(gi >= 1 && gi <= ii && gj >= 1 && gj <= jj)) {
    const int ci = gi < 1 ? 1 : (gi > ii ? ii : gi), cj = gj < 1 ? 1 : (gj > jj ? jj : gj);
    own_halo = (ci - 1) / TI == (int)bx && (cj - 1) / TJ == (int)blockIdx.y;
  }
  // E/W rim cell served by the exchange buffers: strip row r = gj - (1 - 3), column q within the strip
  const bool from_buf = !PERSIST && a.rim_on && inarr && (gi < 1 || gi > ii);
  const double *rb = gi < 1 ? a.rim_w : a.rim_e;
  const bool rb_has = gi < 1 ? a.rim_has_w != 0 : a.rim_has_e != 0;
  const size_t rb_i = from_buf ? (size_t)(gj + HB - 1) * HB + (size_t)(gi < 1 ? gi + HB - 1 : gi - ii - 1) : 0;
  // arctic patch, persistent form: the fold also rewrites cells of the tile itself -- the seam row jj (on the p- and u-grid all
  // of it, on the v-grid its second half) takes the mirror image at every halo update, i.e. before every odd substep
  const bool seam_pb = PERSIST && mine && cs != c, seam_ub = PERSIST && mine && cs_u != c, seam_vb = PERSIST && mine && cs_v != c;
  auto load_state = [&](bool rim_only) {
    if (!act) return;
    const double *g_pb = b_pb[src], *g_ub = b_ub[src], *g_vb = b_vb[src];
    if (rim_only && mine) {
      if (seam_pb || seam_ub || seam_vb) {
#pragma unroll
        for (int l = 0; l < 2; l++) {
          if (seam_pb) s_pb[l][lj][li] = g_pb[cs + l * np];
          if (seam_ub) s_ub[l][lj][li] = sg_u * g_ub[cs_u + l * np];
          if (seam_vb) s_vb[l][lj][li] = sg_v * g_vb[cs_v + l * np];
        }
      }
      return;
    }
    if (from_buf) {
#pragma unroll
      for (int l = 0; l < 2; l++) {
        const size_t o = (size_t)l * a.rim_per + rb_i, fs = (size_t)2 * a.rim_per;
        s_pb[l][lj][li] = rb_has ? rb[o] : V.P.vland;
        s_ub[l][lj][li] = rb_has ? rb[fs + o] : V.P.vland;
        s_vb[l][lj][li] = rb_has ? rb[2 * fs + o] : V.P.vland;
      }
      return;
    }
#pragma unroll
    for (int l = 0; l < 2; l++) {
      s_pb[l][lj][li] = ok_src ? g_pb[cs + l * np] : (inarr ? V.P.vland : 0.);
      s_ub[l][lj][li] = ok_src ? sg_u * g_ub[cs_u + l * np] : (inarr ? V.P.vland : 0.);
      s_vb[l][lj][li] = ok_src ? sg_v * g_vb[cs_v + l * np] : (inarr ? V.P.vland : 0.);
    }
  };
  load_state(false);
  if (act) {
    s_pvo[lj][li] = inarr ? V.f[F_pvtrop_o][c] : 0.;
    s_pvm[lj][li] = inarr ? V.f[F_pvtrop][c + om] : 0.;
    s_pvn[lj][li] = inarr ? V.f[F_pvtrop][c + on] : 0.;
    s_sx[lj][li] = inarr ? V.f[F_scvxi][c] : 0.;
    s_sy[lj][li] = inarr ? V.f[F_scuyi][c] : 0.;
  }
  const bool wp = inarr && V.m[I_ip][c], wu = inarr && V.m[I_iu][c], wv = inarr && V.m[I_iv][c];
  const double wbaro = V.P.wbaro, dlt = V.P.dlt;
  // per-point coefficients (read once, used by both substeps)
  double scp2i = 0.;
  double u_pgo = 0., u_xpo = 0., u_xmo = 0., u_pgm = 0., u_xpm = 0., u_xmm = 0., u_pgn = 0., u_xpn = 0., u_xmn = 0.;
  double u_scuxi = 0., u_scuy = 0., u_tot = 0., u_glue = 0., u_max = 0., u_min = 0.;
  double v_pgo = 0., v_xpo = 0., v_xmo = 0., v_pgm = 0., v_xpm = 0., v_xmm = 0., v_pgn = 0., v_xpn = 0., v_xmn = 0.;
  double v_scvyi = 0., v_scvx = 0., v_tot = 0., v_glue = 0., v_max = 0., v_min = 0.;
  if (wp) scp2i = V.f[F_scp2i][c];
  if (wu) {
    u_pgo = V.f[F_pgfxm_o][c]; u_xpo = V.f[F_xixp_o][c]; u_xmo = V.f[F_xixm_o][c];
    u_pgm = V.f[F_pgfxm][c + om]; u_xpm = V.f[F_xixp][c + om]; u_xmm = V.f[F_xixm][c + om];
    u_pgn = V.f[F_pgfxm][c + on]; u_xpn = V.f[F_xixp][c + on]; u_xmn = V.f[F_xixm][c + on];
    u_scuxi = V.f[F_scuxi][c]; u_scuy = V.f[F_scuy][c]; u_tot = V.f[F_utotn][c]; u_glue = V.f[F_uglue][c];
    u_max = V.f[F_umaxb][c]; u_min = V.f[F_uminb][c];
  }
  if (wv) {
    v_pgo = V.f[F_pgfym_o][c]; v_xpo = V.f[F_xiyp_o][c]; v_xmo = V.f[F_xiym_o][c];
    v_pgm = V.f[F_pgfym][c + om]; v_xpm = V.f[F_xiyp][c + om]; v_xmm = V.f[F_xiym][c + om];
    v_pgn = V.f[F_pgfym][c + on]; v_xpn = V.f[F_xiyp][c + on]; v_xmn = V.f[F_xiym][c + on];
    v_scvyi = V.f[F_scvyi][c]; v_scvx = V.f[F_scvx][c]; v_tot = V.f[F_vtotn][c]; v_glue = V.f[F_vglue][c];
    v_max = V.f[F_vmaxb][c]; v_min = V.f[F_vminb][c];
  }
  double us_acc = 0., uc_acc = 0., vs_acc = 0., vc_acc = 0.;     // ubflxs_t, ubcors_t, vbflxs_t, vbcors_t increments
  if (mine) {
    if (wu) { us_acc = V.f[F_ubflxs_t][c]; uc_acc = V.f[F_ubcors_t][c]; }
    if (wv) { vs_acc = V.f[F_vbflxs_t][c]; vc_acc = V.f[F_vbcors_t][c]; }
  }
  PROF_MARK();
  __syncthreads();
  PROF_MARK();

  int ml = a.ml - 1, nl = a.nl - 1;     // 0-based LDS level indices
  const bool mom_scon = V.P.mommth == 0;
  // persistent form: neighbour tiles (wrapped where the domain is periodic, none where it is closed)
  const int nbx = gridDim.x, nby = gridDim.y;
  int nb_tile = -1;                     // lanes 0..7 of wave 0 watch one neighbour each
  if (PERSIST && tid < 8) {
    const int dx = (tid < 3 ? -1 : (tid < 5 ? 0 : 1)), dy = (tid == 0 || tid == 3 || tid == 5) ? -1 : ((tid == 1 || tid == 6) ? 0 : 1);
    // tid: 0 (-1,-1) 1 (-1,0) 2 (-1,1) 3 (0,-1) 4 (0,1) 5 (1,-1) 6 (1,0) 7 (1,1)
    int qx = (int)bx + dx, qy = (int)blockIdx.y + dy;
    bool exists = true;
    if (qx < 0 || qx >= nbx) { if (V.nreg == 0 || V.nreg == 4) exists = false; else qx = (qx + nbx) % nbx; }
    if (qy < 0 || qy >= nby) { if (V.nreg <= 2) exists = false; else qy = (qy + nby) % nby; }
    if (exists) nb_tile = qy * nbx + qx;
  }
  if (PERSIST && V.nreg == 2 && tid >= 8 && tid < 12 && (int)blockIdx.y == nby - 1) {
    // arctic patch: the rows above the seam mirror the top rows of the tiles that hold the mirrored columns (the tile's own
    // columns and rim, reflected: column i <-> ii+1-i, one more on the u-grid); lanes 8..11 watch up to four of them
    const int lo = ii + 1 - ((int)bx * TI + TI + HB) - 1, hi = ii + 1 - ((int)bx * TI + 1 - HB) + 1;
    // four sample columns (hi - lo + 2) / 3 apart: every tile wider than that (bt_phase_shape) that reaches into [lo, hi]
    // holds one of them
    int col = lo + (tid - 8) * ((hi - lo + 2) / 3);
    if (col > hi) col = hi;
    col = ((col - 1) % ii + ii) % ii + 1;                    // periodic in i
    nb_tile = (nby - 1) * nbx + (col - 1) / TI;
  }
  unsigned done_iters = 0;              // iterations this tile has completed
  int lll = a.lll0;
  bool aborted = false;

#define IN(lo_i, hi_i, lo_j, hi_j) (act && li >= (lo_i) && li <= (hi_i) && lj >= (lo_j) && lj <= (hi_j))

  do {
  int do_odd = a.do_odd, do_even = a.do_even;
  double wo_[2] = {a.wo[0], a.wo[1]}, wm_[2] = {a.wm[0], a.wm[1]}, wn_[2] = {a.wn[0], a.wn[1]};
  if (PERSIST) {                        // the host loop of st_barotp (:352-360, :387-392), per iteration
    const bool odd = lll % 2 == 1, both = odd && lll + 1 <= a.last;
    for (int x = 0; x < 2; x++) {
      const int l = both ? lll + x : lll;
      wo_[x] = a.woa * l + a.wob;
      wn_[x] = a.wna * l + a.wnb;
      wm_[x] = 1. - wo_[x] - wn_[x];
    }
    if (!both && !odd) { wo_[1] = wo_[0]; wm_[1] = wm_[0]; wn_[1] = wn_[0]; }
    do_odd = odd ? 1 : 0;
    do_even = (both || !odd) ? 1 : 0;
    lll += both ? 2 : 1;
    PMARK(0);
    if (done_iters > 0) {               // wait for the neighbours' previous iteration, then re-read the rim
      if (tid < 64) {
        bool ready = nb_tile < 0;
        unsigned spins = 0;
        while (true) {
          if (!ready) ready = __hip_atomic_load(a.flags + nb_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.epoch_base + done_iters;
          if (__all(ready)) break;
          if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { aborted = true; break; }
          if (++spins > 400000u) { __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        WAIT_VMCNT0();
        PMARK(1);
      }
      // one LDS word tells the other waves about an abort
      if (tid == 0) s_abort = aborted ? 1 : 0;
      __syncthreads();
      if (s_abort) break;
      load_state(true);
      __syncthreads();
      PMARK(2);
    }
  }
  // tile-local validity rectangle of what has been computed so far (inclusive, in li/lj)
  int vlo_i = 0, vhi_i = BI - 1, vlo_j = 0, vhi_j = BJ - 1;
  for (int half = 0; half < 2; half++) {
    if (half == 0 ? !do_odd : !do_even) continue;
    const bool odd = half == 0;
    const double wo = wo_[half], wm = wm_[half], wn = wn_[half];
    // ---- continuity: needs ub(i+1), vb(j+1) at level ml ------------------------------------------
    {
      const int r_i0 = odd ? -1 : 0, r_i1 = odd ? ii + 1 : ii, r_j0 = odd ? -1 : 0, r_j1 = odd ? jj + 2 : jj + 1;
      if (wp && IN(vlo_i, vhi_i - 1, vlo_j, vhi_j - 1) && gi >= r_i0 && gi <= r_i1 && gj >= r_j0 && gj <= r_j1)
        s_pb[nl][lj][li] = (1. - wbaro) * s_pb[ml][lj][li] + wbaro * s_pb[nl][lj][li] -
                           (1. + wbaro) * dlt *
                               (s_ub[ml][lj][li + 1] - s_ub[ml][lj][li] + s_vb[ml][lj + 1][li] - s_vb[ml][lj][li]) * scp2i;
    }
    __syncthreads();
    PROF_MARK();
    if (half == 1) PMARK(7);                 // the even substep's continuity sweep done
    // after continuity pb[nl] is valid on [vlo_i, vhi_i-1] x [vlo_j, vhi_j-1]
    const int p_hi_i = vhi_i - 1, p_hi_j = vhi_j - 1;
    auto do_u = [&](int lv, int lo_i, int hi_i, int lo_j, int hi_j, int r_i0, int r_i1, int r_j0, int r_j1) {
      if (wu && IN(lo_i, hi_i, lo_j, hi_j) && gi >= r_i0 && gi <= r_i1 && gj >= r_j0 && gj <= r_j1) {
        const double ubml = s_ub[ml][lj][li], ubnl = s_ub[nl][lj][li];
        if (mine) us_acc = us_acc - wbaro * ubnl + (1. + wbaro) * ubml;
        const double vc = s_vb[lv][lj][li], vn = s_vb[lv][lj + 1][li], vw = s_vb[lv][lj][li - 1], vnw = s_vb[lv][lj + 1][li - 1];
        const double sx_c = s_sx[lj][li], sx_n = s_sx[lj + 1][li], sx_w = s_sx[lj][li - 1], sx_nw = s_sx[lj + 1][li - 1];
        const double pvo_c = s_pvo[lj][li], pvm_c = s_pvm[lj][li], pvn_c = s_pvn[lj][li];
        const double pvo_n = s_pvo[lj + 1][li], pvm_n = s_pvm[lj + 1][li], pvn_n = s_pvn[lj + 1][li];
        double q;
        if (mom_scon)
          q = (vc * sx_c + vn * sx_n + vw * sx_w + vnw * sx_nw) *
              (wo * (pvo_c + pvo_n) + wm * (pvm_c + pvm_n) + wn * (pvn_c + pvn_n)) * .125;
        else
          q = .25 * ((vc * sx_c + vw * sx_w) * (wo * pvo_c + wm * pvm_c + wn * pvn_c) +
                     (vn * sx_n + vnw * sx_nw) * (wo * pvo_n + wm * pvm_n + wn * pvn_n));
        if (mine) uc_acc = uc_acc + q;
        const double pbc = s_pb[nl][lj][li], pbw = s_pb[nl][lj][li - 1];
        const double utndcy = q + (wo * (u_pgo - (u_xpo * pbc - u_xmo * pbw)) + wm * (u_pgm - (u_xpm * pbc - u_xmm * pbw)) +
                                   wn * (u_pgn - (u_xpn * pbc - u_xmn * pbw))) * u_scuxi;
        const double x = (1. - wbaro) * ubml + wbaro * ubnl +
                         (1. + wbaro) * dlt * ((utndcy + u_tot) * u_scuy * fmin2(pbw, pbc) - u_glue * ubml);
        s_ub[nl][lj][li] = fmax2(-u_min, fmin2(u_max, x));
      }
    };
    auto do_v = [&](int lu, int lo_i, int hi_i, int lo_j, int hi_j, int r_i0, int r_i1, int r_j0, int r_j1) {
      if (wv && IN(lo_i, hi_i, lo_j, hi_j) && gi >= r_i0 && gi <= r_i1 && gj >= r_j0 && gj <= r_j1) {
        const double vbml = s_vb[ml][lj][li], vbnl = s_vb[nl][lj][li];
        if (mine) vs_acc = vs_acc - wbaro * vbnl + (1. + wbaro) * vbml;
        const double uc = s_ub[lu][lj][li], ue = s_ub[lu][lj][li + 1], us = s_ub[lu][lj - 1][li], use = s_ub[lu][lj - 1][li + 1];
        const double sy_c = s_sy[lj][li], sy_e = s_sy[lj][li + 1], sy_s = s_sy[lj - 1][li], sy_se = s_sy[lj - 1][li + 1];
        const double pvo_c = s_pvo[lj][li], pvm_c = s_pvm[lj][li], pvn_c = s_pvn[lj][li];
        const double pvo_e = s_pvo[lj][li + 1], pvm_e = s_pvm[lj][li + 1], pvn_e = s_pvn[lj][li + 1];
        double q;
        if (mom_scon)
          q = -(uc * sy_c + ue * sy_e + us * sy_s + use * sy_se) *
              (wo * (pvo_c + pvo_e) + wm * (pvm_c + pvm_e) + wn * (pvn_c + pvn_e)) * .125;
        else
          q = -.25 * ((uc * sy_c + us * sy_s) * (wo * pvo_c + wm * pvm_c + wn * pvn_c) +
                      (ue * sy_e + use * sy_se) * (wo * pvo_e + wm * pvm_e + wn * pvn_e));
        if (mine) vc_acc = vc_acc + q;
        const double pbc = s_pb[nl][lj][li], pbs = s_pb[nl][lj - 1][li];
        const double vtndcy = q + (wo * (v_pgo - (v_xpo * pbc - v_xmo * pbs)) + wm * (v_pgm - (v_xpm * pbc - v_xmm * pbs)) +
                                   wn * (v_pgn - (v_xpn * pbc - v_xmn * pbs))) * v_scvyi;
        const double x = (1. - wbaro) * vbml + wbaro * vbnl +
                         (1. + wbaro) * dlt * ((vtndcy + v_tot) * v_scvx * fmin2(pbs, pbc) - v_glue * vbml);
        s_vb[nl][lj][li] = fmax2(-v_min, fmin2(v_max, x));
      }
    };
    if (odd) {
      // u: needs pb[nl] at i-1,i ; vb[ml] at (i-1..i, j..j+1)                       (:420-457)
      const int u_lo_i = vlo_i + 1, u_hi_i = p_hi_i, u_lo_j = vlo_j, u_hi_j = p_hi_j;
      do_u(ml, u_lo_i, u_hi_i, u_lo_j, u_hi_j, 0, ii + 1, -1, jj + 2);
      __syncthreads();
      // v: needs ub[nl] at (i..i+1, j-1..j) ; pb[nl] at j-1,j                          (:520-557)
      const int v_lo_i = u_lo_i, v_hi_i = u_hi_i - 1, v_lo_j = u_lo_j + 1, v_hi_j = u_hi_j;
      do_v(nl, v_lo_i, v_hi_i, v_lo_j, v_hi_j, 0, ii, 0, jj + 2);
      __syncthreads();
      vlo_i = v_lo_i; vhi_i = v_hi_i; vlo_j = v_lo_j; vhi_j = v_hi_j;
    } else {
      // v first: needs ub[ml] at (i..i+1, j-1..j) ; pb[nl] at j-1,j                     (:646-682)
      const int v_lo_i = vlo_i, v_hi_i = p_hi_i, v_lo_j = vlo_j + 1, v_hi_j = p_hi_j;
      do_v(ml, v_lo_i, v_hi_i, v_lo_j, v_hi_j, 0, ii, 1, jj + 1);
      __syncthreads();
      // u: needs vb[nl] at (i-1..i, j..j+1) ; pb[nl] at i-1,i                           (:745-781)
      const int u_lo_i = v_lo_i + 1, u_hi_i = v_hi_i, u_lo_j = v_lo_j, u_hi_j = v_hi_j - 1;
      do_u(nl, u_lo_i, u_hi_i, u_lo_j, u_hi_j, 1, ii, 1, jj);
      __syncthreads();
      vlo_i = u_lo_i; vhi_i = u_hi_i; vlo_j = u_lo_j; vhi_j = u_hi_j;
    }
    PROF_MARK();
    if (half == 0) PMARK(6);                 // the odd substep's three sweeps done
    const int t = ml; ml = nl; nl = t;       // :614-616 / :837-839
  }
  // publish the interior (and, with the arctic patch, the owned margin) in the other buffer set
  PMARK(3);
  src ^= 1;
  // persistent form: between iterations only the cells within HB of the tile edge are read by anybody (the
  // neighbours' rims); the core of the tile lives in LDS and goes to memory with the last iteration
  const bool edge_cell = li < 2 * HB || li >= TI || lj < 2 * HB || lj >= TJ || gi > ii - HB || gj > jj - HB ||
                         (V.nreg == 2 && gj > jj - 2 * HB);
  if ((mine && (!PERSIST || edge_cell || lll > a.last)) || (own_halo && (!PERSIST || lll > a.last))) {
    double *o_pb = b_pb[src], *o_ub = b_ub[src], *o_vb = b_vb[src];
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if (PERSIST) {                     // write-through stores: visible to the other XCDs without a release fence
        __hip_atomic_store(o_pb + c + l * np, s_pb[l][lj][li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o_ub + c + l * np, s_ub[l][lj][li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o_vb + c + l * np, s_vb[l][lj][li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        o_pb[c + l * np] = s_pb[l][lj][li];
        o_ub[c + l * np] = s_ub[l][lj][li];
        o_vb[c + l * np] = s_vb[l][lj][li];
      }
    }
  }
  if (!PERSIST && a.pack_on && act) {
    // send strips of the NEXT exchange, layout [field][level][row][q] with rows 1-HB..jj+HB (comm_rccl.hip):
    // west strip = columns 1..HB, east strip = columns ii-HB+1..ii
    const bool inw = gi >= 1 && gi <= HB, ine = gi > ii - HB && gi <= ii;
    const bool perj = V.nreg > 2;
    if (inw || ine) {
      const size_t per = (size_t)a.rim_per, fs = 2 * per;
      auto put = [&](int row, bool landv) {                // row = strip row index of global row gj'
        const size_t base = (size_t)row * HB;
#pragma unroll
        for (int l = 0; l < 2; l++) {
          const double vp = landv ? V.P.vland : s_pb[l][lj][li], vu = landv ? V.P.vland : s_ub[l][lj][li];
          const double vv = landv ? V.P.vland : s_vb[l][lj][li];
          if (inw) {
            const size_t o = l * per + base + (size_t)(gi - 1);
            a.pack_w[o] = vp; a.pack_w[fs + o] = vu; a.pack_w[2 * fs + o] = vv;
          }
          if (ine) {
            const size_t o = l * per + base + (size_t)(gi - (ii - HB + 1));
            a.pack_e[o] = vp; a.pack_e[fs + o] = vu; a.pack_e[2 * fs + o] = vv;
          }
        }
      };
      if (mine) {
        put(gj + HB - 1, false);
        if (perj) {                                       // the rows that are the periodic images of halo rows
          if (gj <= HB) put(gj + jj + HB - 1, false);
          if (gj > jj - HB) put(gj - jj + HB - 1, false);
        }
      } else if (!perj && (gi - 1) / TI == (int)bx &&
                 ((gj < 1 && gj >= 1 - HB && blockIdx.y == 0) || (gj > jj && gj <= jj + HB && blockIdx.y == gridDim.y - 1))) {
        put(gj + HB - 1, true);                           // closed in j: halo rows of the strips are land
      }
    }
  }
  if (PERSIST) {
    WAIT_VMCNT0();     // every storing wave drains before the count goes out
    __syncthreads();
    PMARK(4);
    done_iters++;
    if (tid == 0) __hip_atomic_store(a.flags + (blockIdx.y * nbx + bx), a.epoch_base + done_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef BT_PROFILE
    if (PERSIST && a.prof && tid == 0 && done_iters <= 16u)
      a.prof[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + done_iters - 1) * 8 + 5] = wall_clock64();
#endif
  }
  } while (PERSIST && lll <= a.last);
  if (mine) {
    if (wu) { V.f[F_ubflxs_t][c] = us_acc; V.f[F_ubcors_t][c] = uc_acc; }
    if (wv) { V.f[F_vbflxs_t][c] = vs_acc; V.f[F_vbcors_t][c] = vc_acc; }
  }
  PROF_MARK();
}

// ---------------------------------------------------------------------------------------------------------------------
// Temporal blocking: up to FOUR substeps per hand-off.
//
// In k_bt_steps<true> an iteration (one odd+even pair) is 4.3 us of sweeps and 5.5 us of hand-off (stores drained, count
// published, the neighbours' counts seen, rim re-read): the launch is bound by the latency of 63 hand-offs per step, not by
// arithmetic.  A substep consumes at most one cell of valid rim on the low side and two on the high side (continuity reads
// ub(i+1), vb(j+1); the momentum equation that comes second reads the first one's result at i+1 / j+1 again; each reads pb at
// i-1 / j-1), an odd+even pair two and three -- so with a rim of 4 cells below and 6 above the tile, 36 x 26 points around
// 26 x 16, FOUR consecutive substeps of either parity run between two hand-offs: 35 instead of 65 iterations per step.
// The rim cells are computed redundantly, and they are what the neighbour computes for them: the same expressions on the
// same inputs (the reference relies on the same property between its odd and its even substep, which it runs without a
// halo update on margins computed redundantly, phy/mod_barotp.F90:387-843).  For that to hold at the edges of the domain
// every point of the block is identified with its HOME point -- the interior point it is the periodic image of; beyond a
// closed edge: land -- and coefficients, masks and state all come from there (the arrays' halos are 4 wide and hold some of
// the coefficients 2 deep only, :271-285); the loop ranges of the reference's substeps, which differ between odd and even
// substeps by how far they reach into the margins, need no restating then: a margin point is computed whenever its inputs
// are valid, and is the image of an interior point that is.  936 points are two per thread for 512 threads: 8 wavefronts,
// two per SIMD, up to 256 VGPRs -- the 31 coefficients of both points stay in registers.  Not for the arctic patch (whose
// halo update rewrites an interior row), not for the tiles of several processes: k_bt_steps serves those.
// Launched with a range of at most four substeps the kernel is the non-persistent form (no waiting: option barotp_block = 2,
// which is how the host emulation runs it).
#define BT4_RL 4
#define BT4_RH 6
#define BT4_NT 512
struct Bt4Pt {
  int li, lj;
  bool act, mine, wp, wu, wv, land;
  size_t c;
  double scp2i;
  double u_pgo, u_xpo, u_xmo, u_pgm, u_xpm, u_xmm, u_pgn, u_xpn, u_xmn, u_scuxi, u_scuy, u_tot, u_glue, u_max, u_min;
  double v_pgo, v_xpo, v_xmo, v_pgm, v_xpm, v_xmm, v_pgn, v_xpn, v_xmn, v_scvyi, v_scvx, v_tot, v_glue, v_max, v_min;
  double us_acc, uc_acc, vs_acc, vc_acc;
};

template <int TI, int TJ>
__global__ void __launch_bounds__(BT4_NT) k_bt_steps4(const DevView *__restrict__ Vp, PairArgs a) {
  constexpr int RL = BT4_RL, RH = BT4_RH, BI = TI + RL + RH, BJ = TJ + RL + RH, NPT = BI * BJ, NT = BT4_NT;
  static_assert(NPT <= 2 * NT, "two points per thread");
  const DevView &V = *Vp;
  __shared__ double s_pb[2][BJ][BI + 1], s_ub[2][BJ][BI + 1], s_vb[2][BJ][BI + 1];
  __shared__ double s_pvo[BJ][BI + 1], s_pvm[BJ][BI + 1], s_pvn[BJ][BI + 1], s_sx[BJ][BI + 1], s_sy[BJ][BI + 1];
  __shared__ int s_abort;
  const int tid = threadIdx.x;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int ii = V.ii, jj = V.jj;
  const size_t np = V.nplane;
  const gd_t b_pb[2] = {V.f[F_pb_t], V.f[F_pb_t2]}, b_ub[2] = {V.f[F_ubflx_t], V.f[F_ubflx_t2]};
  const gd_t b_vb[2] = {V.f[F_vbflx_t], V.f[F_vbflx_t2]};
  int src = a.src;
  const size_t om = (size_t)(a.m - 1) * np, on = (size_t)(a.n - 1) * np;
  const bool per_i = !(V.nreg == 0 || V.nreg == 4), per_j = V.nreg > 2;
  const double vland = V.P.vland;
  // does the launch start with the first substep of a phase?  (then the phase's flux sums start at zero; else they continue from memory)
  const bool fresh = a.lll0 == 1 || a.lll0 == a.ph_last[0] + 1 || a.lll0 == a.ph_last[1] + 1 || a.lll0 == a.ph_last[2] + 1 || a.lll0 == a.ph_last[3] + 1;
  Bt4Pt P[2];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    Bt4Pt &q = P[p];
    const int x = tid + p * NT;
    q.act = x < NPT;
    q.li = q.act ? x % BI : 0; q.lj = q.act ? x / BI : 0;
    const int gi = bx * TI + 1 + q.li - RL, gj = by * TJ + 1 + q.lj - RL;
    const bool oi = gi < 1 || gi > ii, oj = gj < 1 || gj > jj;
    q.land = !q.act || (oi && !per_i) || (oj && !per_j);
    const int hi = ((gi - 1) % ii + ii) % ii + 1, hj = ((gj - 1) % jj + jj) % jj + 1;
    q.c = q.land ? 0 : (size_t)IDX(V, hi, hj);
    q.mine = q.act && !oi && !oj && q.li >= RL && q.li < RL + TI && q.lj >= RL && q.lj < RL + TJ;
    q.wp = !q.land && V.m[I_ip][q.c]; q.wu = !q.land && V.m[I_iu][q.c]; q.wv = !q.land && V.m[I_iv][q.c];
    const size_t c = q.c;
    if (q.act) {
      s_pvo[q.lj][q.li] = q.land ? 0. : V.f[F_pvtrop_o][c];
      s_pvm[q.lj][q.li] = q.land ? 0. : V.f[F_pvtrop][c + om];
      s_pvn[q.lj][q.li] = q.land ? 0. : V.f[F_pvtrop][c + on];
      s_sx[q.lj][q.li] = q.land ? 0. : V.f[F_scvxi][c];
      s_sy[q.lj][q.li] = q.land ? 0. : V.f[F_scuyi][c];
    }
    q.scp2i = 0.;
    q.u_pgo = q.u_xpo = q.u_xmo = q.u_pgm = q.u_xpm = q.u_xmm = q.u_pgn = q.u_xpn = q.u_xmn = 0.;
    q.u_scuxi = q.u_scuy = q.u_tot = q.u_glue = q.u_max = q.u_min = 0.;
    q.v_pgo = q.v_xpo = q.v_xmo = q.v_pgm = q.v_xpm = q.v_xmm = q.v_pgn = q.v_xpn = q.v_xmn = 0.;
    q.v_scvyi = q.v_scvx = q.v_tot = q.v_glue = q.v_max = q.v_min = 0.;
    q.us_acc = q.uc_acc = q.vs_acc = q.vc_acc = 0.;
    if (q.wp) q.scp2i = V.f[F_scp2i][c];
    if (q.wu) {
      q.u_pgo = V.f[F_pgfxm_o][c]; q.u_xpo = V.f[F_xixp_o][c]; q.u_xmo = V.f[F_xixm_o][c];
      q.u_pgm = V.f[F_pgfxm][c + om]; q.u_xpm = V.f[F_xixp][c + om]; q.u_xmm = V.f[F_xixm][c + om];
      q.u_pgn = V.f[F_pgfxm][c + on]; q.u_xpn = V.f[F_xixp][c + on]; q.u_xmn = V.f[F_xixm][c + on];
      q.u_scuxi = V.f[F_scuxi][c]; q.u_scuy = V.f[F_scuy][c]; q.u_tot = V.f[F_utotn][c]; q.u_glue = V.f[F_uglue][c];
      q.u_max = V.f[F_umaxb][c]; q.u_min = V.f[F_uminb][c];
    }
    if (q.wv) {
      q.v_pgo = V.f[F_pgfym_o][c]; q.v_xpo = V.f[F_xiyp_o][c]; q.v_xmo = V.f[F_xiym_o][c];
      q.v_pgm = V.f[F_pgfym][c + om]; q.v_xpm = V.f[F_xiyp][c + om]; q.v_xmm = V.f[F_xiym][c + om];
      q.v_pgn = V.f[F_pgfym][c + on]; q.v_xpn = V.f[F_xiyp][c + on]; q.v_xmn = V.f[F_xiym][c + on];
      q.v_scvyi = V.f[F_scvyi][c]; q.v_scvx = V.f[F_scvx][c]; q.v_tot = V.f[F_vtotn][c]; q.v_glue = V.f[F_vglue][c];
      q.v_max = V.f[F_vmaxb][c]; q.v_min = V.f[F_vminb][c];
    }
    if (q.mine && !fresh) {
      if (q.wu) { q.us_acc = V.f[F_ubflxs_t][c]; q.uc_acc = V.f[F_ubcors_t][c]; }
      if (q.wv) { q.vs_acc = V.f[F_vbflxs_t][c]; q.vc_acc = V.f[F_vbcors_t][c]; }
    }
  }
  auto load_state = [&](bool rim_only) {
    const double *g_pb = b_pb[src], *g_ub = b_ub[src], *g_vb = b_vb[src];
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const Bt4Pt &q = P[p];
      if (!q.act || (rim_only && q.mine)) continue;
#pragma unroll
      for (int l = 0; l < 2; l++) {
        s_pb[l][q.lj][q.li] = q.land ? vland : g_pb[q.c + l * np];
        s_ub[l][q.lj][q.li] = q.land ? vland : g_ub[q.c + l * np];
        s_vb[l][q.lj][q.li] = q.land ? vland : g_vb[q.c + l * np];
      }
    }
  };
  load_state(false);
  __syncthreads();

  int ml = a.ml - 1, nl = a.nl - 1;
  const double wbaro = V.P.wbaro, dlt = V.P.dlt;
  const bool mom_scon = V.P.mommth == 0;
  const int nbx = gridDim.x, nby = gridDim.y;
  int nb_tile = -1;                     // lanes 0..7 of wave 0 watch one neighbour each
  if (tid < 8) {
    const int dx = (tid < 3 ? -1 : (tid < 5 ? 0 : 1)), dy = (tid == 0 || tid == 3 || tid == 5) ? -1 : ((tid == 1 || tid == 6) ? 0 : 1);
    int qx = bx + dx, qy = by + dy;
    bool exists = true;
    if (qx < 0 || qx >= nbx) { if (!per_i) exists = false; else qx = (qx + nbx) % nbx; }
    if (qy < 0 || qy >= nby) { if (!per_j) exists = false; else qy = (qy + nby) % nby; }
    if (exists) nb_tile = qy * nbx + qx;
  }
  unsigned done_iters = 0;
  int lll = a.lll0;
  bool aborted = false, ended_phase = false;

  do {
    const int nsub = a.last - lll + 1 < 4 ? a.last - lll + 1 : 4;
    if (done_iters > 0) {               // wait for the neighbours' previous iteration, then re-read the rim
      if (tid < 64) {
        bool ready = nb_tile < 0;
        unsigned spins = 0;
        while (true) {
          if (!ready) ready = __hip_atomic_load(a.flags + nb_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.epoch_base + done_iters;
          if (__all(ready)) break;
          if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { aborted = true; break; }
          if (++spins > 400000u) { __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        WAIT_VMCNT0();
      }
      if (tid == 0) s_abort = aborted ? 1 : 0;
      __syncthreads();
      if (s_abort) break;
      load_state(true);
      __syncthreads();
    }
    // block-local rectangle of what is valid so far (inclusive)
    int vlo_i = 0, vhi_i = BI - 1, vlo_j = 0, vhi_j = BJ - 1;
    for (int s = 0; s < nsub; s++) {
      const int l = lll + s;
      const bool odd = l % 2 == 1;
      int ip = 0;                                                                                // the phase of substep l
      while (ip < 4 && l > a.ph_last[ip]) ip++;
      if (l == (ip ? a.ph_last[ip - 1] + 1 : 1)) {                                               // :361-379
#pragma unroll
        for (int p = 0; p < 2; p++) P[p].us_acc = P[p].uc_acc = P[p].vs_acc = P[p].vc_acc = 0.;
      }
      const double wo = a.ph_w[ip][0] * l + a.ph_w[ip][1], wn = a.ph_w[ip][2] * l + a.ph_w[ip][3], wm = 1. - wo - wn;        // :352-360
      // ---- continuity (:400-418 / :625-643): reads ub(i+1), vb(j+1) at level ml
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const Bt4Pt &q = P[p];
        const int li = q.li, lj = q.lj;
        if (q.wp && li >= vlo_i && li <= vhi_i - 1 && lj >= vlo_j && lj <= vhi_j - 1)
          s_pb[nl][lj][li] = (1. - wbaro) * s_pb[ml][lj][li] + wbaro * s_pb[nl][lj][li] -
                             (1. + wbaro) * dlt *
                                 (s_ub[ml][lj][li + 1] - s_ub[ml][lj][li] + s_vb[ml][lj + 1][li] - s_vb[ml][lj][li]) * q.scp2i;
      }
      __syncthreads();
      const int p_hi_i = vhi_i - 1, p_hi_j = vhi_j - 1;
      auto do_u = [&](Bt4Pt &q, int lv, int lo_i, int hi_i, int lo_j, int hi_j) {
        const int li = q.li, lj = q.lj;
        if (q.wu && li >= lo_i && li <= hi_i && lj >= lo_j && lj <= hi_j) {
          const double ubml = s_ub[ml][lj][li], ubnl = s_ub[nl][lj][li];
          if (q.mine) q.us_acc = q.us_acc - wbaro * ubnl + (1. + wbaro) * ubml;
          const double vc = s_vb[lv][lj][li], vn = s_vb[lv][lj + 1][li], vw = s_vb[lv][lj][li - 1], vnw = s_vb[lv][lj + 1][li - 1];
          const double sx_c = s_sx[lj][li], sx_n = s_sx[lj + 1][li], sx_w = s_sx[lj][li - 1], sx_nw = s_sx[lj + 1][li - 1];
          const double pvo_c = s_pvo[lj][li], pvm_c = s_pvm[lj][li], pvn_c = s_pvn[lj][li];
          const double pvo_n = s_pvo[lj + 1][li], pvm_n = s_pvm[lj + 1][li], pvn_n = s_pvn[lj + 1][li];
          double r;
          if (mom_scon)
            r = (vc * sx_c + vn * sx_n + vw * sx_w + vnw * sx_nw) *
                (wo * (pvo_c + pvo_n) + wm * (pvm_c + pvm_n) + wn * (pvn_c + pvn_n)) * .125;
          else
            r = .25 * ((vc * sx_c + vw * sx_w) * (wo * pvo_c + wm * pvm_c + wn * pvn_c) +
                       (vn * sx_n + vnw * sx_nw) * (wo * pvo_n + wm * pvm_n + wn * pvn_n));
          if (q.mine) q.uc_acc = q.uc_acc + r;
          const double pbc = s_pb[nl][lj][li], pbw = s_pb[nl][lj][li - 1];
          const double utndcy = r + (wo * (q.u_pgo - (q.u_xpo * pbc - q.u_xmo * pbw)) + wm * (q.u_pgm - (q.u_xpm * pbc - q.u_xmm * pbw)) +
                                     wn * (q.u_pgn - (q.u_xpn * pbc - q.u_xmn * pbw))) * q.u_scuxi;
          const double x = (1. - wbaro) * ubml + wbaro * ubnl +
                           (1. + wbaro) * dlt * ((utndcy + q.u_tot) * q.u_scuy * fmin2(pbw, pbc) - q.u_glue * ubml);
          s_ub[nl][lj][li] = fmax2(-q.u_min, fmin2(q.u_max, x));
        }
      };
      auto do_v = [&](Bt4Pt &q, int lu, int lo_i, int hi_i, int lo_j, int hi_j) {
        const int li = q.li, lj = q.lj;
        if (q.wv && li >= lo_i && li <= hi_i && lj >= lo_j && lj <= hi_j) {
          const double vbml = s_vb[ml][lj][li], vbnl = s_vb[nl][lj][li];
          if (q.mine) q.vs_acc = q.vs_acc - wbaro * vbnl + (1. + wbaro) * vbml;
          const double uc = s_ub[lu][lj][li], ue = s_ub[lu][lj][li + 1], us = s_ub[lu][lj - 1][li], use = s_ub[lu][lj - 1][li + 1];
          const double sy_c = s_sy[lj][li], sy_e = s_sy[lj][li + 1], sy_s = s_sy[lj - 1][li], sy_se = s_sy[lj - 1][li + 1];
          const double pvo_c = s_pvo[lj][li], pvm_c = s_pvm[lj][li], pvn_c = s_pvn[lj][li];
          const double pvo_e = s_pvo[lj][li + 1], pvm_e = s_pvm[lj][li + 1], pvn_e = s_pvn[lj][li + 1];
          double r;
          if (mom_scon)
            r = -(uc * sy_c + ue * sy_e + us * sy_s + use * sy_se) *
                (wo * (pvo_c + pvo_e) + wm * (pvm_c + pvm_e) + wn * (pvn_c + pvn_e)) * .125;
          else
            r = -.25 * ((uc * sy_c + us * sy_s) * (wo * pvo_c + wm * pvm_c + wn * pvn_c) +
                        (ue * sy_e + use * sy_se) * (wo * pvo_e + wm * pvm_e + wn * pvn_e));
          if (q.mine) q.vc_acc = q.vc_acc + r;
          const double pbc = s_pb[nl][lj][li], pbs = s_pb[nl][lj - 1][li];
          const double vtndcy = r + (wo * (q.v_pgo - (q.v_xpo * pbc - q.v_xmo * pbs)) + wm * (q.v_pgm - (q.v_xpm * pbc - q.v_xmm * pbs)) +
                                     wn * (q.v_pgn - (q.v_xpn * pbc - q.v_xmn * pbs))) * q.v_scvyi;
          const double x = (1. - wbaro) * vbml + wbaro * vbnl +
                           (1. + wbaro) * dlt * ((vtndcy + q.v_tot) * q.v_scvx * fmin2(pbs, pbc) - q.v_glue * vbml);
          s_vb[nl][lj][li] = fmax2(-q.v_min, fmin2(q.v_max, x));
        }
      };
      if (odd) {
        // u (:420-457): pb[nl] at i-1, i; vb[ml] at (i-1..i, j..j+1).  Then v (:520-557): ub[nl] at (i..i+1, j-1..j); pb[nl] at j-1, j
        const int u_lo_i = vlo_i + 1, u_hi_i = p_hi_i, u_lo_j = vlo_j, u_hi_j = p_hi_j;
#pragma unroll
        for (int p = 0; p < 2; p++) do_u(P[p], ml, u_lo_i, u_hi_i, u_lo_j, u_hi_j);
        __syncthreads();
        const int v_lo_i = u_lo_i, v_hi_i = u_hi_i - 1, v_lo_j = u_lo_j + 1, v_hi_j = u_hi_j;
#pragma unroll
        for (int p = 0; p < 2; p++) do_v(P[p], nl, v_lo_i, v_hi_i, v_lo_j, v_hi_j);
        __syncthreads();
        vlo_i = v_lo_i; vhi_i = v_hi_i; vlo_j = v_lo_j; vhi_j = v_hi_j;
      } else {
        // v first (:646-682): ub[ml] at (i..i+1, j-1..j); pb[nl] at j-1, j.  Then u (:745-781): vb[nl] at (i-1..i, j..j+1); pb[nl] at i-1, i
        const int v_lo_i = vlo_i, v_hi_i = p_hi_i, v_lo_j = vlo_j + 1, v_hi_j = p_hi_j;
#pragma unroll
        for (int p = 0; p < 2; p++) do_v(P[p], ml, v_lo_i, v_hi_i, v_lo_j, v_hi_j);
        __syncthreads();
        const int u_lo_i = v_lo_i + 1, u_hi_i = v_hi_i, u_lo_j = v_lo_j, u_hi_j = v_hi_j - 1;
#pragma unroll
        for (int p = 0; p < 2; p++) do_u(P[p], nl, u_lo_i, u_hi_i, u_lo_j, u_hi_j);
        __syncthreads();
        vlo_i = u_lo_i; vhi_i = u_hi_i; vlo_j = u_lo_j; vhi_j = u_hi_j;
      }
      const int t = ml; ml = nl; nl = t;       // :614-616 / :837-839
      ended_phase = l == a.ph_last[ip];
      if (ended_phase) {                       // the phase's last substep: its flux sums and its epilogue, for the tile's own points
      #pragma unroll
        for (int p = 0; p < 2; p++) {
          const Bt4Pt &q = P[p];
          if (!q.mine) continue;
          const size_t c = q.c;
          if (q.wu) { V.f[F_ubflxs_t][c] = q.us_acc; V.f[F_ubcors_t][c] = q.uc_acc; }
          if (q.wv) { V.f[F_vbflxs_t][c] = q.vs_acc; V.f[F_vbcors_t][c] = q.vc_acc; }
          // the phase's epilogue (k_bt_epilogue of stage_barotp.hip, phy/mod_barotp.F90:847-977) for this point, from the block in LDS: pb at
          // i-1 and j-1 lies in the rim (valid one cell further down than the fluxes after any substep) and is what a halo update would bring
          const int nb = ip + 1, li = q.li, lj = q.lj;
          const size_t oml = (size_t)ml * np, onl = (size_t)nl * np, o3 = 2 * np;
          const double us = q.us_acc, vs = q.vs_acc;
          const double pbc = s_pb[ml][lj][li];
          if (nb == 1 || nb == 3) {
            const size_t ol = nb == 1 ? om : on;
            if (q.wp) V.f[F_pb][c + ol] = pbc;
            if (q.wu) {
              const double pbu = fmin2(pbc, s_pb[ml][lj][li - 1]);
              V.f[F_pbu][c + ol] = pbu;
              const double f = s_ub[ml][lj][li];
              V.f[F_ubflx][c + ol] = f;
              V.f[F_ub][c + ol] = f / (pbu * q.u_scuy);
              if (nb == 1) {
                V.f[F_ubflxs][c + on] = V.f[F_ubflxs][c + on] + us;
                V.f[F_ubflxs][c + om] = V.f[F_ubflxs][c + o3] + us;
              } else {
                V.f[F_ubflxs_p][c + om] = V.f[F_ubflxs][c + om] + us;
                V.f[F_ubflxs_p][c + on] = V.f[F_ubflxs_p][c + on] + us;
                V.f[F_ubcors_p][c] = V.f[F_ubcors_p][c] + q.uc_acc;
              }
            }
            if (q.wv) {
              const double pbv = fmin2(pbc, s_pb[ml][lj - 1][li]);
              V.f[F_pbv][c + ol] = pbv;
              const double f = s_vb[ml][lj][li];
              V.f[F_vbflx][c + ol] = f;
              V.f[F_vb][c + ol] = f / (pbv * q.v_scvx);
              if (nb == 1) {
                V.f[F_vbflxs][c + on] = V.f[F_vbflxs][c + on] + vs;
                V.f[F_vbflxs][c + om] = V.f[F_vbflxs][c + o3] + vs;
              } else {
                V.f[F_vbflxs_p][c + om] = V.f[F_vbflxs][c + om] + vs;
                V.f[F_vbflxs_p][c + on] = V.f[F_vbflxs_p][c + on] + vs;
                V.f[F_vbcors_p][c] = V.f[F_vbcors_p][c] + q.vc_acc;
              }
            }
          } else if (nb == 2) {
            if (q.wp) { V.f[F_pb_mn][c + oml] = pbc; V.f[F_pb_mn][c + onl] = s_pb[nl][lj][li]; }
            if (q.wu) {
              V.f[F_ubflx_mn][c + oml] = s_ub[ml][lj][li];
              V.f[F_ubflx_mn][c + onl] = s_ub[nl][lj][li];
              V.f[F_ubflxs][c + om] = V.f[F_ubflxs][c + om] + us;
              V.f[F_ubflxs][c + o3] = us;
              V.f[F_ubflxs_p][c + on] = us;
              V.f[F_ubcors_p][c] = q.uc_acc;
            }
            if (q.wv) {
              V.f[F_vbflx_mn][c + oml] = s_vb[ml][lj][li];
              V.f[F_vbflx_mn][c + onl] = s_vb[nl][lj][li];
              V.f[F_vbflxs][c + om] = V.f[F_vbflxs][c + om] + vs;
              V.f[F_vbflxs][c + o3] = vs;
              V.f[F_vbflxs_p][c + on] = vs;
              V.f[F_vbcors_p][c] = q.vc_acc;
            }
          } else {
            if (nb == 5) {
              if (q.wp) V.f[F_pb_p][c] = pbc;
              if (q.wu) V.f[F_pbu_p][c] = fmin2(pbc, s_pb[ml][lj][li - 1]);
              if (q.wv) V.f[F_pbv_p][c] = fmin2(pbc, s_pb[ml][lj - 1][li]);
            }
            if (q.wu) {
              V.f[F_ubflxs_p][c + on] = V.f[F_ubflxs_p][c + on] + us;
              V.f[F_ubcors_p][c] = V.f[F_ubcors_p][c] + q.uc_acc;
            }
            if (q.wv) {
              V.f[F_vbflxs_p][c + on] = V.f[F_vbflxs_p][c + on] + vs;
              V.f[F_vbcors_p][c] = V.f[F_vbcors_p][c] + q.vc_acc;
            }
          }
        }
      }
    }
    lll += nsub;
    // publish the tile in the other buffer set (between iterations the neighbours read all but 14 x 4 of its points)
    src ^= 1;
    {
      double *o_pb = b_pb[src], *o_ub = b_ub[src], *o_vb = b_vb[src];
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const Bt4Pt &q = P[p];
        if (!q.mine) continue;
#pragma unroll
        for (int l = 0; l < 2; l++) {        // write-through stores: visible to the other XCDs without a release fence
          __hip_atomic_store(o_pb + q.c + l * np, s_pb[l][q.lj][q.li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(o_ub + q.c + l * np, s_ub[l][q.lj][q.li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(o_vb + q.c + l * np, s_vb[l][q.lj][q.li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (lll > a.last) break;
    WAIT_VMCNT0();     // every storing wave drains before the count goes out
    __syncthreads();
    done_iters++;
    if (tid == 0) __hip_atomic_store(a.flags + (by * nbx + bx), a.epoch_base + done_iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } while (true);
  if (!ended_phase) {                  // (a launch that stops inside a phase -- barotp_block = 2 -- leaves the sums for the next one)
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const Bt4Pt &q = P[p];
      if (!q.mine) continue;
      if (q.wu) { V.f[F_ubflxs_t][q.c] = q.us_acc; V.f[F_ubcors_t][q.c] = q.uc_acc; }
      if (q.wv) { V.f[F_vbflxs_t][q.c] = q.vs_acc; V.f[F_vbcors_t][q.c] = q.vc_acc; }
    }
  }
}

// Halo update of the three subcycling fields of buffer set `set`, both levels, in ONE launch:
// widths (3,3), a superset of the reference's (2,2),(2,2),(2,3) at :395-397.  Same gather rule as
// k_xctilr_single (halo.hip): closed direction -> vland, periodic direction -> wrapped source.
__global__ void k_bt_halo3(const DevView *__restrict__ Vp, int set, int mhl, int nhl) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj;
  const int nns = 2 * nhl * ii, new_ = 2 * mhl * (jj + 2 * nhl);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nns + new_) return;
  int i, j;
  if (t < nns) {
    const int r = t / ii;
    i = t % ii + 1;
    j = r < nhl ? -r : jj + (r - nhl) + 1;
  } else {
    t -= nns;
    const int cidx = t % (2 * mhl);
    j = t / (2 * mhl) + 1 - nhl;
    i = cidx < mhl ? -cidx : ii + (cidx - mhl) + 1;
  }
  const bool inew = i < 1 || i > ii, ins = j < 1 || j > jj;
  const bool land = (inew && (V.nreg == 0 || V.nreg == 4)) || (ins && V.nreg <= 2);
  int is = i, js = j;
  if (i < 1) is = i + ii; else if (i > ii) is = i - ii;
  if (j < 1) js = j + jj; else if (j > jj) js = j - jj;
  const size_t dst = IDX(V, i, j), src = IDX(V, is, js);
  double *f = blockIdx.y == 0 ? (set ? V.f[F_pb_t2] : V.f[F_pb_t])
            : blockIdx.y == 1 ? (set ? V.f[F_ubflx_t2] : V.f[F_ubflx_t]) : (set ? V.f[F_vbflx_t2] : V.f[F_vbflx_t]);
  for (int l = 0; l < 2; l++) {
    double *pl = f + (size_t)l * V.nplane;
    pl[dst] = land ? V.P.vland : pl[src];
  }
}

// RCCL tiles: exchange without the unpack launch; the strips stay in the transport's receive buffers
int bt_pair_halo_landed(blomgpu_ctx *c, int set, RcclLanded *landed) {
  const DevView &h = c->h;
  double *f[3] = {set ? h.f[F_pb_t2] : h.f[F_pb_t], set ? h.f[F_ubflx_t2] : h.f[F_ubflx_t],
                  set ? h.f[F_vbflx_t2] : h.f[F_vbflx_t]};
  return rccl_xctilr_multi_ex(c, f, 3, 2, 3, 3, landed);
}

int bt_pair_halo(blomgpu_ctx *c, int set) {
  const DevView &h = c->h;
  if (c->tiling.multi() || h.nreg == 2) {   // neighbour exchange through the tile transport / the arctic rule of xctilr
    double *f[3] = {set ? h.f[F_pb_t2] : h.f[F_pb_t], set ? h.f[F_ubflx_t2] : h.f[F_ubflx_t],
                    set ? h.f[F_vbflx_t2] : h.f[F_vbflx_t]};
    static const int it[3] = {1, 13, 14};
    if (h.nreg == 2) {                                                     // single tile, RCCL (batched), tiles of one process
      const int nl3[3] = {2, 2, 2}, w3[3] = {3, 3, 3};
      return st_xctilr_arctic_multi(c, 3, f, nl3, w3, w3, it);
    }
    if (c->tiling.rccl) return rccl_xctilr_multi(c, f, 3, 2, 3, 3);       // one message per neighbour
    for (int x = 0; x < 3; x++)
      if (int rc = st_xctilr(c, f[x], 1, 2, 3, 3, it[x])) return rc;
    return 0;
  }
  const int ntarget = 2 * 3 * h.ii + 2 * 3 * (h.jj + 6);
  hipLaunchKernelGGL(k_bt_halo3, dim3((ntarget + 255) / 256, 3), dim3(256), 0, c->stream, c->d, set, 3, 3);
  return 0;
}

bool bt_phase_usable(blomgpu_ctx *c);
// Tile shape of the one-pair-per-launch form.  Where the persistent form runs, its shape; otherwise chosen from the
// number of tiles (option barotp_tile = 3216 / 3208 / 1608 overrides).
static BtShape bt_phase_shape(blomgpu_ctx *c);
static BtShape bt_shape(blomgpu_ctx *c) {
  if (c->barotp_tile) return BtShape{c->barotp_tile / 100, c->barotp_tile % 100};
  const DevView &h = c->h;
  if (c->barotp_persist && bt_phase_usable(c)) return bt_phase_shape(c);
  if (c->num_cus <= 0) {
    hipDeviceProp_t prop;
    c->num_cus = hipGetDeviceProperties(&prop, c->device) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  // the smallest shape whose tiles all find a CU of their own (measured on the 180 x 193 tripolar grid: 32x16 = 78 tiles
  // 16.4 us per launch, 32x8 = 150 tiles 13.5 us, 16x8 = 300 tiles on 256 CUs 22.7 us)
  // 40x16 (1012 of the 1024 threads a workgroup may have): for domains whose 32x16 tiles outnumber the CUs -- the tnx1v4
  // dimensions, 360 x 385: 300 tiles of 32x16 take two rounds on 256 CUs (42 us per pair), 225 tiles of 40x16 one
  static const BtShape shapes[5] = {{16, 8}, {32, 8}, {26, 16}, {32, 16}, {40, 16}};
  for (const BtShape &sh : shapes) {
    const int nt = ((h.ii + sh.ti - 1) / sh.ti) * ((h.jj + sh.tj - 1) / sh.tj);
    if (nt <= c->num_cus) return sh;
  }
  return shapes[3];
}
static int bt_launch_pair(blomgpu_ctx *c, BtShape sh, dim3 grid, hipStream_t st, const PairArgs &a) {
  if (sh.ti == 32 && sh.tj == 16) hipLaunchKernelGGL((k_bt_steps<false, 32, 16>), grid, dim3(bt_threads(32, 16)), 0, st, c->d, a);
  else if (sh.ti == 32 && sh.tj == 8) hipLaunchKernelGGL((k_bt_steps<false, 32, 8>), grid, dim3(bt_threads(32, 8)), 0, st, c->d, a);
  else if (sh.ti == 16 && sh.tj == 8) hipLaunchKernelGGL((k_bt_steps<false, 16, 8>), grid, dim3(bt_threads(16, 8)), 0, st, c->d, a);
  else if (sh.ti == 40 && sh.tj == 16) hipLaunchKernelGGL((k_bt_steps<false, 40, 16>), grid, dim3(bt_threads(40, 16)), 0, st, c->d, a);
  else if (sh.ti == 26 && sh.tj == 16) hipLaunchKernelGGL((k_bt_steps<false, 26, 16>), grid, dim3(bt_threads(26, 16)), 0, st, c->d, a);
  else if (sh.ti == 26 && sh.tj == 15) hipLaunchKernelGGL((k_bt_steps<false, 26, 15>), grid, dim3(bt_threads(26, 15)), 0, st, c->d, a);
  else if (sh.ti == 40 && sh.tj == 15) hipLaunchKernelGGL((k_bt_steps<false, 40, 15>), grid, dim3(bt_threads(40, 15)), 0, st, c->d, a);
  else return ctx_fail(c, "barotp: tile shape must be 3216, 3208, 1608, 4016, 2616, 2615 or 4015");
  return 0;
}

int bt_pair_launch(blomgpu_ctx *c, int m, int n, int ml, int nl, const double *wo, const double *wm, const double *wn,
                   int do_odd, int do_even, int src, int tsel, RcclLanded *rim) {
  const DevView &h = c->h;
  PairArgs a;
  a.m = m; a.n = n; a.ml = ml; a.nl = nl;
  for (int x = 0; x < 2; x++) { a.wo[x] = wo[x]; a.wm[x] = wm[x]; a.wn[x] = wn[x]; }
  a.do_odd = do_odd; a.do_even = do_even; a.src = src;
  // single tile: the halo rule is applied while loading; with the arctic patch only where the reference has its halo
  // update, in front of an odd substep (a lone even substep reads the margins the launch before it published)
  a.fold_halo = c->tiling.multi() ? 0 : (h.nreg == 2 ? (do_odd ? 1 : 0) : 1);
  a.write_margin = (h.nreg == 2 && (!c->tiling.multi() || c->barotp_arctic_fused)) ? 1 : 0;
  a.rim_on = 0; a.rim_w = a.rim_e = nullptr; a.rim_has_w = a.rim_has_e = 0; a.rim_per = 0;
  if (rim && rim->from_west) {
    a.rim_on = 1; a.rim_w = rim->from_west; a.rim_e = rim->from_east;
    a.rim_has_w = rim->has_w; a.rim_has_e = rim->has_e; a.rim_per = rim->per;
  }
  a.pack_on = 0; a.pack_w = a.pack_e = nullptr;
  if (rim && rim->from_west && rim->send_west && h.jj >= 2 * HB && h.ii >= 2 * HB) {
    a.pack_on = 1; a.pack_w = rim->send_west; a.pack_e = rim->send_east;
  }
  a.prof = c->bt_prof;
  a.lll0 = a.last = 0; a.woa = a.wob = a.wna = a.wnb = 0.; a.flags = nullptr; a.abort_word = nullptr; a.epoch_base = 0;
  const BtShape sh = bt_shape(c);
  const int nbx = (h.ii + sh.ti - 1) / sh.ti, nby = (h.jj + sh.tj - 1) / sh.tj;
  a.nbx = nbx;
  if (rim) rim->prepacked = a.pack_on;      // the launch below leaves the next exchange's send strips packed
  if (tsel == 0) {
    a.tsel = 0;
    return bt_launch_pair(c, sh, dim3(nbx, nby), c->stream, a);
  }
  // edge tile columns on the exchange stream (their output is what gets packed), the rest on the main stream
  a.tsel = 1;
  if (int rc = bt_launch_pair(c, sh, dim3(nbx > 1 ? 2 : 1, nby), c->xstream, a)) return rc;
  if (nbx > 2) {
    a.tsel = 2;
    if (int rc = bt_launch_pair(c, sh, dim3(nbx - 2, nby), c->stream, a)) return rc;
  }
  return 0;
}

// RCCL tiles along i, closed in j: the E/W strips a pair produces come from the two outer columns of
// LDS tiles only, so those run first on the exchange stream, followed by pack / ncclSend+Recv / unpack,
// while the columns in between compute on the main stream (the reference's "exchange, then compute"
// order, phy/mod_barotp.F90:395-397, with the exchange moved to where its input is ready).  With a
// periodic j the N/S phase of the exchange would read rows the inner tiles are still writing.
// MEASURED (channel, RCCL self-send, one MI355X): bit-identical, but barotp 4.87 ms instead of 3.26 ms.
// The pair kernel is latency-bound (15 of its 20 us are the coefficient load), so two half-size
// launches each still take ~20 us, and the cross-stream fork/join events add ~10 us per pair: there is
// no throughput-bound interior to hide the exchange behind.  Kept as an option (barotp_overlap=1),
// off by default.
int bt_overlap_usable(blomgpu_ctx *c) {
  const DevView &h = c->h;
  if (!c->tiling.rccl || !c->xstream || !c->barotp_overlap) return 0;
  if (h.nreg > 2 || h.nreg == 2) return 0;
  if (c->tiling.npy != 1) return 0;
  const BtShape sh = bt_shape(c);
  return (h.ii + sh.ti - 1) / sh.ti >= 3 ? 1 : 0;
}

// Can the persistent form be used, and with which tile shape?  Every tile must be resident (one workgroup per CU) and a
// tile's rim must come from its direct neighbours only (also across the periodic seam; with the arctic patch from the
// mirror tiles of the last tile row too).  Shapes in the order of preference: 26 x 16 and 26 x 15 stay below 768 threads
// (168 VGPRs: no spills); 40 x 16 / 40 x 15 serve domains that need more than 256 of those (the tnx1v4 dimensions).
// {0, 0}: not usable -- one launch per substep pair then.
struct BtPersistShape { int ti, tj; };
static const BtPersistShape kPersistShapes[4] = {{26, 16}, {26, 15}, {40, 16}, {40, 15}};
template <int TI, int TJ>
static int bt_persist_blocks_per_cu() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bt_steps<true, TI, TJ>, bt_threads(TI, TJ), 0) != hipSuccess) nb = 0;
  return nb;
}
static BtShape bt_phase_shape(blomgpu_ctx *c) {
  const DevView &h = c->h;
  const BtShape none{0, 0};
  if (c->tiling.multi()) return none;
  if (c->num_cus <= 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return none;
    c->num_cus = prop.multiProcessorCount;
  }
  for (int x = 0; x < 4; x++) {
    const int ti = kPersistShapes[x].ti, tj = kPersistShapes[x].tj;
    if (c->barotp_tile && c->barotp_tile != 100 * ti + tj) continue;
    const int nbx = (h.ii + ti - 1) / ti, nby = (h.jj + tj - 1) / tj;
    if (nbx * nby > c->num_cus) continue;
    if (h.ii - (nbx - 1) * ti < HB || h.jj - (nby - 1) * tj < HB) continue;
    // arctic patch: the fold reads the rows jj-1-HB..jj, which must lie in the last tile row; at least two tile columns,
    // the last one wider than the sampling step with which the kernel finds its mirror tiles
    if (h.nreg == 2 && (h.jj - (nby - 1) * tj < 2 * HB || nbx < 2 || h.ii - (nbx - 1) * ti < (ti + 2 * HB + 3) / 3 + 1 || !c->barotp_arctic_persist)) continue;
    // the tiles wait for each other inside the launch: every one of them must be resident.  Ask the runtime how many
    // workgroups of this kernel a CU takes (registers, LDS) instead of assuming one.
    if (c->bt_blocks_per_cu[x] < 0)
      c->bt_blocks_per_cu[x] = x == 0 ? bt_persist_blocks_per_cu<26, 16>() : x == 1 ? bt_persist_blocks_per_cu<26, 15>()
                               : x == 2 ? bt_persist_blocks_per_cu<40, 16>() : bt_persist_blocks_per_cu<40, 15>();
    if (c->bt_blocks_per_cu[x] < 1) continue;
    return BtShape{ti, tj};
  }
  return none;
}
bool bt_phase_usable(blomgpu_ctx *c) { return bt_phase_shape(c).ti != 0; }

// one launch for the substeps lll0..last of a phase; `src` as in bt_pair_launch; returns the buffer set
// that holds the state afterwards in *src_out and the level indices in *ml_out, *nl_out
int bt_phase_launch(blomgpu_ctx *c, int m, int n, int ml, int nl, double woa, double wob, double wna, double wnb, int lll0,
                    int last, int src, int *src_out, int *ml_out, int *nl_out) {
  const DevView &h = c->h;
  const BtShape psh = bt_phase_shape(c);
  if (!psh.ti) return ctx_fail(c, "barotp: the persistent form is not usable on this domain");
  const int nbx = (h.ii + psh.ti - 1) / psh.ti, nby = (h.jj + psh.tj - 1) / psh.tj;
  int niter = 0;
  for (int lll = lll0; lll <= last; niter++) lll += (lll % 2 == 1 && lll + 1 <= last) ? 2 : 1;
  // the completion counters start at 0 with the first launch of every barotp call (st_barotp sets bt_restart): the
  // launches of a step then carry the same epochs every step, which lets blomgpu_step replay them as a graph
  if (!c->bt_flags || c->bt_restart) {
    if (!c->bt_flags) HIPCHK(c, hipMalloc((void **)&c->bt_flags, sizeof(unsigned) * (nbx * nby + 16)));
    HIPCHK(c, hipMemsetAsync(c->bt_flags, 0, sizeof(unsigned) * (nbx * nby + 16), c->stream));
    c->bt_epoch = 0;
    c->bt_restart = false;
  }
  PairArgs a;
  a.m = m; a.n = n; a.ml = ml; a.nl = nl;
  for (int x = 0; x < 2; x++) { a.wo[x] = a.wm[x] = a.wn[x] = 0.; }
  a.do_odd = a.do_even = 0; a.src = src;
  a.fold_halo = 1;
  a.prof = c->bt_prof;
  a.lll0 = lll0; a.last = last; a.woa = woa; a.wob = wob; a.wna = wna; a.wnb = wnb;
  a.flags = c->bt_flags + 16;
  // arctic patch: the launch covers whole odd+even pairs only (st_barotp_on); its last iteration also publishes the margins
  // beyond the edges of the domain, which a following lone even substep and the epilogue read without a halo update
  a.tsel = 0; a.nbx = nbx; a.write_margin = h.nreg == 2 ? 1 : 0;
  a.pack_on = 0; a.pack_w = a.pack_e = nullptr;
  a.rim_on = 0; a.rim_w = a.rim_e = nullptr; a.rim_has_w = a.rim_has_e = 0; a.rim_per = 0;
  a.epoch_base = c->bt_epoch;
  c->bt_epoch += (unsigned)niter;
  if (int rc = ctx_err_words(c)) return rc;
  a.abort_word = (unsigned *)(c->err_dev + 2);
  TimeScope tk(c, "k_bt_steps");
  if (psh.ti == 26 && psh.tj == 16) hipLaunchKernelGGL((k_bt_steps<true, 26, 16>), dim3(nbx, nby), dim3(bt_threads(26, 16)), 0, c->stream, c->d, a);
  else if (psh.ti == 26) hipLaunchKernelGGL((k_bt_steps<true, 26, 15>), dim3(nbx, nby), dim3(bt_threads(26, 15)), 0, c->stream, c->d, a);
  else if (psh.tj == 16) hipLaunchKernelGGL((k_bt_steps<true, 40, 16>), dim3(nbx, nby), dim3(bt_threads(40, 16)), 0, c->stream, c->d, a);
  else hipLaunchKernelGGL((k_bt_steps<true, 40, 15>), dim3(nbx, nby), dim3(bt_threads(40, 15)), 0, c->stream, c->d, a);
  // replay the iteration bookkeeping of the kernel
  for (int lll = lll0; lll <= last;) {
    const bool odd = lll % 2 == 1, both = odd && lll + 1 <= last;
    src ^= 1;
    if (!both) { const int t = ml; ml = nl; nl = t; }
    lll += both ? 2 : 1;
  }
  *src_out = src; *ml_out = ml; *nl_out = nl;
  return 0;
}

// The temporally blocked form (k_bt_steps4): shape 26 x 16 or 26 x 15, no arctic patch, one process; the last tile row and
// column at least as wide as the high rim (a tile's rim comes from its direct neighbours only).  persistent: all tiles
// resident.  {0, 0}: not usable.
static BtShape bt_block_shape(blomgpu_ctx *c, bool persistent) {
  const DevView &h = c->h;
  const BtShape none{0, 0};
  if (c->tiling.multi() || h.nreg == 2) return none;
  if (persistent && c->num_cus <= 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return none;
    c->num_cus = prop.multiProcessorCount;
  }
  for (int x = 0; x < 2; x++) {
    const int ti = kPersistShapes[x].ti, tj = kPersistShapes[x].tj;
    if (c->barotp_tile && c->barotp_tile != 100 * ti + tj) continue;
    const int nbx = (h.ii + ti - 1) / ti, nby = (h.jj + tj - 1) / tj;
    if (h.ii - (nbx - 1) * ti < BT4_RH || h.jj - (nby - 1) * tj < BT4_RH) continue;
    if (persistent) {
      if (nbx * nby > c->num_cus) continue;
      if (c->bt4_blocks_per_cu[x] < 0) {
        int nb = 0;
        hipError_t e = x == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bt_steps4<26, 16>, BT4_NT, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bt_steps4<26, 15>, BT4_NT, 0);
        c->bt4_blocks_per_cu[x] = e == hipSuccess ? nb : 0;
      }
      if (c->bt4_blocks_per_cu[x] < 1) continue;
    }
    return BtShape{ti, tj};
  }
  return none;
}
// 1: the persistent blocked form serves this context, 2: the blocked form with one launch per iteration, 0: neither
int bt_block_mode(blomgpu_ctx *c) {
  if (!c->barotp_block || !c->barotp_fused) return 0;
  if (c->barotp_block == 1) return c->barotp_persist && bt_block_shape(c, true).ti ? 1 : 0;
  return bt_block_shape(c, false).ti ? 2 : 0;
}
// the substeps of all five phases, four per iteration; mode as bt_block_mode returns it; ph_last / ph_w: PairArgs
int bt_block_launch(blomgpu_ctx *c, int mode, int m, int n, int ml, int nl, const int *ph_last, const double (*ph_w)[4], int src,
                    int *src_out, int *ml_out, int *nl_out) {
  const DevView &h = c->h;
  const BtShape sh = bt_block_shape(c, mode == 1);
  if (!sh.ti) return ctx_fail(c, "barotp: the blocked form is not usable on this domain");
  const int nbx = (h.ii + sh.ti - 1) / sh.ti, nby = (h.jj + sh.tj - 1) / sh.tj;
  const int lll0 = 1, last = ph_last[4];
  const int niter = (last - lll0 + 1 + 3) / 4;
  PairArgs a;
  a.m = m; a.n = n;
  for (int x = 0; x < 2; x++) { a.wo[x] = a.wm[x] = a.wn[x] = 0.; }
  a.do_odd = a.do_even = 0; a.fold_halo = 1; a.prof = nullptr;
  a.woa = a.wob = a.wna = a.wnb = 0.;
  for (int x = 0; x < 5; x++) {
    a.ph_last[x] = ph_last[x];
    for (int y = 0; y < 4; y++) a.ph_w[x][y] = ph_w[x][y];
  }
  a.tsel = 0; a.nbx = nbx; a.write_margin = 0;
  a.pack_on = 0; a.pack_w = a.pack_e = nullptr;
  a.rim_on = 0; a.rim_w = a.rim_e = nullptr; a.rim_has_w = a.rim_has_e = 0; a.rim_per = 0;
  a.flags = nullptr; a.abort_word = nullptr; a.epoch_base = 0;
  auto launch = [&]() {
    if (sh.tj == 16) hipLaunchKernelGGL((k_bt_steps4<26, 16>), dim3(nbx, nby), dim3(BT4_NT), 0, c->stream, c->d, a);
    else hipLaunchKernelGGL((k_bt_steps4<26, 15>), dim3(nbx, nby), dim3(BT4_NT), 0, c->stream, c->d, a);
  };
  if (mode == 1) {
    if (!c->bt_flags || c->bt_restart) {
      if (!c->bt_flags) HIPCHK(c, hipMalloc((void **)&c->bt_flags, sizeof(unsigned) * (nbx * nby + 16)));
      HIPCHK(c, hipMemsetAsync(c->bt_flags, 0, sizeof(unsigned) * (nbx * nby + 16), c->stream));
      c->bt_epoch = 0;
      c->bt_restart = false;
    }
    a.flags = c->bt_flags + 16;
    a.epoch_base = c->bt_epoch;
    c->bt_epoch += (unsigned)niter;
    if (int rc = ctx_err_words(c)) return rc;
    a.abort_word = (unsigned *)(c->err_dev + 2);
    a.ml = ml; a.nl = nl; a.src = src; a.lll0 = lll0; a.last = last;
    TimeScope tk(c, "k_bt_steps");
    launch();
    HIPCHK(c, hipGetLastError());
    src ^= niter & 1;
    if ((last - lll0 + 1) & 1) { const int t = ml; ml = nl; nl = t; }
  } else {
    TimeScope tk(c, "k_bt_steps");
    for (int lll = lll0; lll <= last; lll += 4) {
      const int e = lll + 3 < last ? lll + 3 : last;
      a.ml = ml; a.nl = nl; a.src = src; a.lll0 = lll; a.last = e;
      launch();
      src ^= 1;
      if ((e - lll + 1) & 1) { const int t = ml; ml = nl; nl = t; }
    }
    HIPCHK(c, hipGetLastError());
  }
  *src_out = src; *ml_out = ml; *nl_out = nl;
  return 0;
}

// did any tile of the persistent launches give up waiting?  (checked once per barotp call)
int bt_phase_check(blomgpu_ctx *c) {
  if (c->defer_checks) return 0;
  return ctx_check_errors(c);
}
