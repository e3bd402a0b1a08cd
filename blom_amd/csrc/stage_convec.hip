// convec -- removal of static instabilities between the mixed layer and the interior layers, then
// redistribution of momentum onto the new layer structure (phy/mod_convec.F90:43-449; called between
// momtum and diapfl for isopyc_bulkml, phy/mod_blom_step.F90:172-176).
//
// One thread per column.  The reference copies a column into 1-D arrays, edits them and copies them
// back; the 1-D arrays ARE the column's planes here, edited in place (the planes of 64 neighbouring
// columns are contiguous, so every access of a wavefront is one 512-byte segment).  Almost all columns
// are statically stable and leave after one density comparison; their only work is the interface
// pressure scan p(k+1) = p(k) + dp(k).  The velocity remap needs the old column after the new one has
// started to be written, so it goes through one work plane stack.
#include "blomgpu_internal.h"
#include "eos.h"
#include "diapfl_common.h"

#define PLANE_IJ(V)                                                        \
  unsigned bx_, by_;                                                       \
  xcd_block(bx_, by_);                                                     \
  const int t_ = bx_ * blockDim.x + threadIdx.x;                           \
  (void)by_;                                                               \
  if (t_ >= (V).nplane) return;                                            \
  const int i = t_ % (V).ni - (NBDY - 1), j = t_ / (V).ni - (NBDY - 1);    \
  const size_t c = t_;                                                     \
  (void)i; (void)j; (void)c

#define MAXTR 64   // tracer sums of a column: dynamically indexed (private memory), any tracer count up to this
enum { CV_UN = 0 };   // work field: remapped velocity column

__global__ __launch_bounds__(64) void k_convec_column(const DevView *__restrict__ Vp, int n, int nn, int *errflag) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii || !V.m[I_ip][c]) return;
  const int kk = V.kk, ntr = V.ntr;
  const size_t np = V.nplane;
  const double epsilp = 1.e-12;
  // 1-based level k of the time level n: element c + (k - 1 + nn) * np
  gd_t ttem = V.f[F_temp] + c + (size_t)nn * np - np, ssal = V.f[F_saln] + c + (size_t)nn * np - np;
  gd_t delp = V.f[F_dp] + c + (size_t)nn * np - np, dens = V.f[F_sigma] + c + (size_t)nn * np - np;
  gcd_t densr = V.f[F_sigmar] + c - np;
  gd_t trc = V.f[F_trc] + c + (size_t)nn * np - np; // tracer nt: + nt * 2 * kk * np
  const size_t ntl = (size_t)2 * kk * np;
#define TT(k) ttem[(size_t)(k) * np]
#define SS(k) ssal[(size_t)(k) * np]
#define DP(k) delp[(size_t)(k) * np]
#define DN(k) dens[(size_t)(k) * np]
#define DR(k) densr[(size_t)(k) * np]
#define TR(nt, k) trc[(size_t)(nt) * ntl + (size_t)(k) * np]
  double tdps, sdps, dps, ttmp, stmp, dtmp, q = 0.;

  // first physical interior layer, :95-109
  int k = 3;
  dps = 0.;
  {
    // (the walk over the massless layers under the mixed layer, COLUMN_U levels' loads ahead: with a deep mixed layer it is 20-30
    // levels long, one dependent load each)
    bool walking = true;
    for (int k0 = 3; walking && k0 <= kk; k0 += COLUMN_U) {
      double a[COLUMN_U];
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) a[u] = DP(k0 + u <= kk ? k0 + u : kk);
#pragma unroll
      for (int u = 0; u < COLUMN_U; u++) {
        if (!walking || k0 + u > kk) break;
        if (a[u] < epsilp) {
          dps = dps + a[u];
          DP(k0 + u) = 0.;
          k = k0 + u + 1;
        } else
          walking = false;
      }
    }
  }
  if (k > kk) DP(2) = DP(2) + dps;
  else DP(k) = DP(k) + dps;
  int kfpl = k;
  gi_t kfpla = V.m[I_kfpla] + c + (size_t)(n - 1) * np;
  const int kfplo = *kfpla;
  // The tracers' share of a mixing event, :118-122 / :160-164 / :213-217, :233-237 and the assignments that follow them: each tracer's
  // thickness-weighted sum over the levels ka..kb in the reference's order (after layer 2's term where the event starts there), times
  // q, into one or two levels.  The reference carries the sums along while it looks for the event's extent; they depend on nothing but
  // dp, so they are evaluated here once the extent is known -- and not at all in a column without an event, which is nearly every
  // column: CV_TB tracers at a time, COLUMN_U levels' loads in flight.  (Must run before the event's layers give up their dp.)
#define CV_TB 4
  auto mix_tracers = [&](bool from2, double d2, int ka, int kb, double qq, int kd1, int kd2) {
    for (int nt0 = 0; nt0 < ntr; nt0 += CV_TB) {
      double acc[CV_TB];
#pragma unroll
      for (int b = 0; b < CV_TB; b++) {
        const int nt = nt0 + b < ntr ? nt0 + b : ntr - 1;
        acc[b] = 0.;
        if (from2) acc[b] = TR(nt, 2) * d2;
      }
      for (int k0 = ka; k0 <= kb; k0 += COLUMN_U) {
        double d[COLUMN_U], x[CV_TB][COLUMN_U];
#pragma unroll
        for (int u = 0; u < COLUMN_U; u++) {
          const int kq = k0 + u <= kb ? k0 + u : kb;
          d[u] = DP(kq);
#pragma unroll
          for (int b = 0; b < CV_TB; b++) x[b][u] = TR(nt0 + b < ntr ? nt0 + b : ntr - 1, kq);
        }
#pragma unroll
        for (int u = 0; u < COLUMN_U; u++)
          if (k0 + u <= kb) {
#pragma unroll
            for (int b = 0; b < CV_TB; b++) acc[b] = acc[b] + x[b][u] * d[u];
          }
      }
#pragma unroll
      for (int b = 0; b < CV_TB; b++)
        if (nt0 + b < ntr) {
          const double r = acc[b] * qq;
          TR(nt0 + b, kd1) = r;
          if (kd2) TR(nt0 + b, kd2) = r;
        }
    }
  };
  if (kfpl < kfplo) {                                        // :110-191
    tdps = 0.; sdps = 0.; dps = 0.;
    const int kfpl0 = kfpl;
    if (kfplo <= kk) {
      for (k = kfpl; k <= kfplo; k++) {
        const double d = DP(k);
        tdps = tdps + TT(k) * d;
        sdps = sdps + SS(k) * d;
        dps = dps + d;
      }
      q = 1. / dps;
      ttmp = tdps * q;
      stmp = sdps * q;
      dtmp = eos::sig(V.P, ttmp, stmp);
      if (dtmp > DR(kfplo)) {
        mix_tracers(false, 0., kfpl0, kfplo, q, kfplo, 0);
        for (k = kfpl; k <= kfplo - 1; k++) DP(k) = 0.;
        kfpl = kfplo;
        TT(kfpl) = ttmp; SS(kfpl) = stmp; DN(kfpl) = dtmp; DP(kfpl) = dps;
      }
    } else {
      for (k = kfpl; k <= kk; k++) {
        const double d = DP(k);
        tdps = tdps + TT(k) * d;
        sdps = sdps + SS(k) * d;
        dps = dps + d;
      }
      q = 1. / dps;
      ttmp = tdps * q;
      stmp = sdps * q;
      dtmp = eos::sig(V.P, ttmp, stmp);
      kfpl = kk;
      while (dtmp < DR(kfpl)) {
        if (kfpl == 3) break;
        kfpl = kfpl - 1;
      }
      mix_tracers(false, 0., kfpl0, kk, q, kfpl, 0);
      for (k = kfpl0; k <= kk; k++) DP(k) = 0.;
      TT(kfpl) = ttmp; SS(kfpl) = stmp; DN(kfpl) = dtmp; DP(kfpl) = dps;
    }
  }

  if (kfpl <= kk) {                                          // :193-283
    bool done = false;
    int niter = 0;
    while (!done) {
      niter = niter + 1;
      if (niter == 100) {                                    // the reference prints and goes on, :203-206
        atomicOr(errflag, 1);
        break;
      }
      done = true;
      const double t2 = TT(2), s2 = SS(2), d2 = DP(2);
      tdps = t2 * d2;
      sdps = s2 * d2;
      dps = d2;
      ttmp = t2;
      stmp = s2;
      k = kfpl;
      double tk = TT(k), sk = SS(k), dk = DP(k);
      while (true) {
        const int kn = k + 1 <= kk ? k + 1 : kk;             // (the next level's values are on their way during the test)
        const double tn = TT(kn), sn = SS(kn), dn = DP(kn);
        if (!(eos::rho(dps, ttmp, stmp) > eos::rho(dps, tk, sk) || dk < epsilp)) break;
        tdps = tdps + tk * dk;
        sdps = sdps + sk * dk;
        dps = dps + dk;
        q = 1. / dps;
        ttmp = tdps * q;
        stmp = sdps * q;
        k = k + 1;
        if (k > kk) break;
        tk = tn; sk = sn; dk = dn;
      }
      const int kmix = k - 1;
      if (kmix >= kfpl) {
        const double dn2 = eos::sig(V.P, ttmp, stmp);
        TT(2) = ttmp;
        SS(2) = stmp;
        DN(2) = dn2;
        k = kmix;
        while (dn2 < DR(k)) {
          if (k == 3) break;
          k = k - 1;
        }
        mix_tracers(true, d2, kfpl, kmix, q, 2, k);         // ttrc(nt,2) = trdps(nt)*q, ttrc(nt,kfpl) = ttrc(nt,2)
        dps = 0.;
        for (int kz = kfpl; kz <= kmix; kz++) {
          dps = dps + DP(kz);
          DP(kz) = 0.;
        }
        kfpl = k;
        TT(kfpl) = ttmp; SS(kfpl) = stmp; DN(kfpl) = dn2; DP(kfpl) = dps;
        for (k = kfpl + 1; k <= kmix; k++) {
          const double dr = DR(k);
          TT(k) = ttmp;
          DN(k) = dr;
          SS(k) = eosd::sofsig(V.P, dr, ttmp);
        }
      }
    }
  }
  *kfpla = kfpl;
  // :288-302, COLUMN_U levels' loads in flight (nearly every column is stable and does nothing but this scan: one dependent load per
  // level made the kernel 0.2 ms of waiting, 90 % of its wave cycles parked)
  column_scan(V.f[F_p][c], delp + np, V.f[F_p] + c, np, kk);
#undef TT
#undef SS
#undef DP
#undef DN
#undef DR
#undef TR
}

// :315-391: u (blockIdx.y = 0) / v (1) columns are remapped conservatively from the old interface
// pressures at the velocity point (pu, pv) to the new ones
// NSINGLE: moves of a level walked one by one before the walk switches to chunks of CV_B old layers (A/B: 0 .. 3; 1000 = never)
template <int NSINGLE>
__global__ __launch_bounds__(64) void k_convec_velocity(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!V.m[isv ? I_iv : I_iu][c]) return;
  const int kk = V.kk;
  const size_t np = V.nplane, cm = isv ? c - V.ni : c - 1;
  gd_t vel = V.f[isv ? F_v : F_u] + c + (size_t)nn * np - np; // 1-based level
  gcd_t po = V.f[isv ? F_pv : F_pu] + c - np; // po(k), k = 2..kk+1; po(1) = 0
  gcd_t p = V.f[F_p] - np;
  gd_t un = WK(V, CV_UN + (isv ? 1 : 0)) + c - np;
  const double pbot = po[(size_t)(kk + 1) * np];
  // The old layer ko that holds the new interface moves down with kn, usually by one: its velocity and lower interface
  // are kept in registers together with those of layer ko+1, re-loaded in the background when ko moves; the new
  // interfaces' pressures of CV_U levels are loaded ahead (u is not written before the copy at the end).
#define CV_U 8
#define CV_B 6
  int ko = 1;
  double po_lo = 0., po_hi = po[(size_t)2 * np];                         // po(ko), po(ko+1)
  double v_cur = vel[(size_t)1 * np], v_nxt = vel[(size_t)(kk >= 2 ? 2 : 1) * np];       // u(ko), u(ko+1)
  double po_nx = kk >= 2 ? po[(size_t)3 * np] : 1.e300;                   // po(ko+2)
  double pn_lo = 0.;                                                     // pn(kn)
  for (int k0 = 1; k0 <= kk; k0 += CV_U) {
    double a0[CV_U], a1[CV_U];
#pragma unroll
    for (int u = 0; u < CV_U; u++) {
      const int kq = k0 + u <= kk ? k0 + u : kk;
      a0[u] = p[c + (size_t)(kq + 1) * np]; a1[u] = p[cm + (size_t)(kq + 1) * np];
    }
#pragma unroll
    for (int u = 0; u < CV_U; u++) {
      const int kn = k0 + u;
      if (kn > kk) break;
      const double pn_hi = .5 * (fmin2(pbot, a0[u]) + fmin2(pbot, a1[u]));
      double r;
      if (pn_hi - pn_lo == 0.) {
        r = 0.;
      } else {
        double udpn = 0.;
        // The first two old layers a new one leaves behind are walked with their successor in registers, one load pair per move (the
        // common case: layers of similar thickness).  A new layer that spans a RUN of old ones -- deep mixed layers put dozens of
        // massless layers between two interfaces -- then requests the next CV_B old layers at once and walks them from registers: one
        // memory round trip per CV_B layers instead of one per layer (round 6: 204 -> 158 us after 600 steps of the bench workload).
        // The sum's terms and their order are unchanged.
        int nmv = 0;
        while (pn_hi > po_hi) {
          if (NSINGLE >= 1000 || nmv < NSINGLE) {
            nmv++;
            udpn = udpn + v_cur * (po_hi - fmax2(po_lo, pn_lo));
            ko = ko + 1;
            po_lo = po_hi;
            po_hi = ko <= kk ? po_nx : 1.e300;   // (never reached: pn <= po(kk+1))
            v_cur = v_nxt;
            v_nxt = vel[(size_t)(ko + 1 <= kk ? ko + 1 : kk) * np];
            po_nx = ko + 1 <= kk ? po[(size_t)(ko + 2) * np] : 1.e300;
          } else {
            double w[CV_B], q[CV_B];
#pragma unroll
            for (int b = 0; b < CV_B; b++) {
              const int kv = ko + 2 + b;
              w[b] = vel[(size_t)(kv <= kk ? kv : kk) * np];
              q[b] = po[(size_t)(kv <= kk ? kv + 1 : kk + 1) * np];
            }
#pragma unroll
            for (int b = 0; b < CV_B; b++)
              if (pn_hi > po_hi) {
                udpn = udpn + v_cur * (po_hi - fmax2(po_lo, pn_lo));
                ko = ko + 1;
                po_lo = po_hi;
                po_hi = ko <= kk ? po_nx : 1.e300;
                v_cur = v_nxt;
                v_nxt = w[b];
                po_nx = ko + 1 <= kk ? q[b] : 1.e300;
              }
          }
        }
        r = (udpn + v_cur * (pn_hi - fmax2(po_lo, pn_lo))) / (pn_hi - pn_lo);
      }
      un[(size_t)kn * np] = r;
      pn_lo = pn_hi;
    }
  }
  for (int k0 = 1; k0 <= kk; k0 += CV_U) {
    double a0[CV_U];
#pragma unroll
    for (int u = 0; u < CV_U; u++) a0[u] = un[(size_t)(k0 + u <= kk ? k0 + u : kk) * np];
#pragma unroll
    for (int u = 0; u < CV_U; u++)
      if (k0 + u <= kk) vel[(size_t)(k0 + u) * np] = a0[u];
  }
}

// The same remap as ONE loop (round 6; DESIGN.md 3.4 rule 7).  k_convec_velocity is a merge of two sorted interface lists written as a
// loop nest -- for every new layer, while the old layer ends above the new interface: move on -- and a wavefront pays, for every new
// layer, its slowest lane's moves, each with a memory round trip in its path.  Here a trip does for every lane what that lane does next,
// one move or one finished layer (<= 2 kk trips, none of them waiting for memory): a lane's next CVF_W new interfaces and next CVF_W old
// layers (velocity, lower interface) lie in LDS rings, [slot][lane], which the wave tops up together -- every lane requests what its rings
// have room for, up to 3 CVF_W loads in flight per lane -- whenever one of its lanes is about to run dry.  The sums' terms and order are
// the nested form's.  LDS 9 KB per wavefront: all 3 510 waves stay resident.
#define CVF_W 6
__global__ __launch_bounds__(64) void k_convec_velocity_flat(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  PLANE_IJ(V);
  __shared__ double s_pn[CVF_W][64], s_vel[CVF_W][64], s_po[CVF_W][64];
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!V.m[isv ? I_iv : I_iu][c]) return;
  const int kk = V.kk, ln = threadIdx.x;
  const size_t np = V.nplane, cm = isv ? c - V.ni : c - 1;
  gd_t vel = V.f[isv ? F_v : F_u] + c + (size_t)nn * np - np; // 1-based level
  gcd_t po = V.f[isv ? F_pv : F_pu] + c - np; // po(k), k = 2..kk+1; po(1) = 0
  gcd_t p = V.f[F_p] - np;
  gd_t un = WK(V, CV_UN + (isv ? 1 : 0)) + c - np;
  const double pbot = po[(size_t)(kk + 1) * np];
  // old side: layer ko with its velocity and interfaces in registers; ring entries m = ko+1 .. of-1 hold V(m) = u(min(m, kk)) and
  // Q(m) = po(m+1) (1e300 beyond the bottom: never reached, pn <= po(kk+1)); (v_nx, q_nx) is the entry of m = ko+1 when hv_o
  int ko = 1, of = 2, so_r = 2 % CVF_W, so_w = 2 % CVF_W;       // slots of the entries ko+1 (read) and of (write)
  double po_lo = 0., po_hi = po[(size_t)2 * np], v_cur = vel[(size_t)1 * np], v_nx = 0., q_nx = 0.;
  bool hv_o = false;
  // new side: ring entries k = kn .. nf-1 hold pn(k+1); pn_hi is the entry of kn when hv_n
  int kn = 1, nf = 1, sn_r = 1 % CVF_W, sn_w = 1 % CVF_W;
  double pn_lo = 0., pn_hi = 0., udpn = 0.;
  bool hv_n = false;
  while (kn <= kk) {
    if (__any(!hv_n || !hv_o)) {
      double a0[CVF_W], a1[CVF_W], w[CVF_W], q[CVF_W];
#pragma unroll
      for (int b = 0; b < CVF_W; b++) {
        const int lv = nf + b <= kk ? nf + b : kk, m = of + b <= kk ? of + b : kk;
        a0[b] = p[c + (size_t)(lv + 1) * np]; a1[b] = p[cm + (size_t)(lv + 1) * np];
        w[b] = vel[(size_t)m * np]; q[b] = po[(size_t)(m + 1) * np];
      }
#pragma unroll
      for (int b = 0; b < CVF_W; b++) {
        if (nf <= kk && nf < kn + CVF_W) {                         // room in the ring and a level left
          s_pn[sn_w][ln] = .5 * (fmin2(pbot, a0[b]) + fmin2(pbot, a1[b]));
          nf++; sn_w = sn_w + 1 == CVF_W ? 0 : sn_w + 1;
        }
        if (of <= kk + 1 && of <= ko + CVF_W) {
          s_vel[so_w][ln] = w[b]; s_po[so_w][ln] = of <= kk ? q[b] : 1.e300;
          of++; so_w = so_w + 1 == CVF_W ? 0 : so_w + 1;
        }
      }
    }
    if (!hv_n) { pn_hi = s_pn[sn_r][ln]; hv_n = true; }           // (kn <= kk here, and the top-up has put level kn there)
    if (!hv_o) {
      if (ko + 1 <= kk + 1) { v_nx = s_vel[so_r][ln]; q_nx = s_po[so_r][ln]; }
      else { v_nx = 0.; q_nx = 1.e300; }
      hv_o = true;
    }
    const bool massless = pn_hi - pn_lo == 0.;
    if (!massless && pn_hi > po_hi) {                              // the old layer ends above the new interface: move on
      udpn = udpn + v_cur * (po_hi - fmax2(po_lo, pn_lo));
      ko = ko + 1;
      po_lo = po_hi;
      po_hi = q_nx;
      v_cur = v_nx;
      so_r = so_r + 1 == CVF_W ? 0 : so_r + 1;
      if (ko + 1 > kk + 1) { v_nx = 0.; q_nx = 1.e300; }           // below the bottom: po_hi is 1e300 by now, no further move
      else if (ko + 1 < of) { v_nx = s_vel[so_r][ln]; q_nx = s_po[so_r][ln]; }
      else hv_o = false;
    } else {                                                       // the new layer is complete
      const double r = massless ? 0. : (udpn + v_cur * (pn_hi - fmax2(po_lo, pn_lo))) / (pn_hi - pn_lo);
      un[(size_t)kn * np] = r;
      pn_lo = pn_hi;
      udpn = 0.;
      kn = kn + 1;
      sn_r = sn_r + 1 == CVF_W ? 0 : sn_r + 1;
      if (kn <= kk) {
        if (kn < nf) pn_hi = s_pn[sn_r][ln];
        else hv_n = false;
      }
    }
  }
  for (int k0 = 1; k0 <= kk; k0 += CV_U) {
    double a0[CV_U];
#pragma unroll
    for (int u = 0; u < CV_U; u++) a0[u] = un[(size_t)(k0 + u <= kk ? k0 + u : kk) * np];
#pragma unroll
    for (int u = 0; u < CV_U; u++)
      if (k0 + u <= kk) vel[(size_t)(k0 + u) * np] = a0[u];
  }
}

// :393-414
__global__ void k_convec_dpudpv(const DevView *__restrict__ Vp, int nn) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t_ = bx_ * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const size_t c = t_;
  const int k = by_;
  const size_t np = V.nplane, o0 = (size_t)k * np, o1 = (size_t)(k + 1) * np, ob = (size_t)V.kk * np;
  gcd_t p = V.f[F_p];
  if (V.m[I_iu][c]) {
    const size_t w = c - 1;
    const double q = fmin2(p[c + ob], p[w + ob]);
    V.f[F_dpu][c + (size_t)(k + nn) * np] =
        .5 * ((fmin2(q, p[w + o1]) - fmin2(q, p[w + o0])) + (fmin2(q, p[c + o1]) - fmin2(q, p[c + o0])));
  }
  if (V.m[I_iv][c]) {
    const size_t s = c - V.ni;
    const double q = fmin2(p[c + ob], p[s + ob]);
    V.f[F_dpv][c + (size_t)(k + nn) * np] =
        .5 * ((fmin2(q, p[s + o1]) - fmin2(q, p[s + o0])) + (fmin2(q, p[c + o1]) - fmin2(q, p[c + o0])));
  }
}

// the velocity remap on its own: mxlayr ends with the same one (phy/mod_mxlayr.F90:1312-1374)
int st_convec_velocity(blomgpu_ctx *c, int nn) {
  switch (c->convec_nsingle) {
    case -1: hipLaunchKernelGGL(k_convec_velocity_flat, plane_grid(c->h, 2, 64), dim3(64), 0, c->stream, c->d, nn); break;
    case 0: hipLaunchKernelGGL(k_convec_velocity<0>, plane_grid(c->h, 2, 64), dim3(64), 0, c->stream, c->d, nn); break;
    case 1: hipLaunchKernelGGL(k_convec_velocity<1>, plane_grid(c->h, 2, 64), dim3(64), 0, c->stream, c->d, nn); break;
    case 2: hipLaunchKernelGGL(k_convec_velocity<2>, plane_grid(c->h, 2, 64), dim3(64), 0, c->stream, c->d, nn); break;
    case 3: hipLaunchKernelGGL(k_convec_velocity<3>, plane_grid(c->h, 2, 64), dim3(64), 0, c->stream, c->d, nn); break;
    default: hipLaunchKernelGGL(k_convec_velocity<1000>, plane_grid(c->h, 2, 64), dim3(64), 0, c->stream, c->d, nn); break;
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// inside blomgpu_step: the column kernel on the second stream, beside momtum (called at the start of st_momtum, i.e. after
// pgforc, the last reader of dp, T, S of level n in front of it; momtum reads none of what the kernel writes -- its p is its own copy)
int st_convec_column_ahead(blomgpu_ctx *c, int n, int nn) {
  const DevView &h = c->h;
  c->convec_col_ahead = false;
  if (h.ntr > MAXTR || h.P.vcoord_tag != 1) return 0;       // st_convec will refuse
  if (int rc = ctx_err_words(c)) return rc;
  if (int rc = ctx_side_fork(c, 2)) return rc;
  hipLaunchKernelGGL(k_convec_column, plane_grid(h, 1, 64), dim3(64), 0, c->side, c->d, n, nn, c->err_dev + 3);
  HIPCHK(c, hipGetLastError());
  if (int rc = ctx_side_done(c, 3)) return rc;
  c->convec_col_ahead = true;
  return 0;
}

int st_convec(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  (void)m; (void)mm; (void)k1m; (void)k1n;
  const DevView &h = c->h;
  if (h.ntr > MAXTR) return ctx_fail(c, "convec: too many tracers for the device kernels");
  if (h.P.vcoord_tag != 1) return ctx_fail(c, "convec is only called for isopyc_bulkml (phy/mod_blom_step.F90:172-176)");
  if (int rc = ctx_err_words(c)) return rc;
  {
    TimeScope ts(c, "convec");
    if (c->convec_col_ahead) {
      c->convec_col_ahead = false;
      if (int rc = ctx_side_join(c, 3)) return rc;
    } else
      hipLaunchKernelGGL(k_convec_column, plane_grid(h, 1, 64), dim3(64), 0, c->stream, c->d, n, nn, c->err_dev + 3);
    if (int rc = st_xctilr(c, h.f[F_p], 1, h.kk + 1, 1, 1, 1)) return rc;                       // :313
    if (int rc = st_convec_velocity(c, nn)) return rc;
    hipLaunchKernelGGL(k_convec_dpudpv, plane_grid(h, h.kk), dim3(256), 0, c->stream, c->d, nn);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
