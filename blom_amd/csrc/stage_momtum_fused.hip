// momtum, layer loop as two row-marching fused kernels -- phy/mod_momtum.F90:342-1144.
//
// The one-kernel-per-sweep form (stage_momtum.hip) sends ~30 kk-level temporaries through HBM (7.5 GB per call for
// 1.2 GB of algorithmic bytes).  Here a workgroup spans a whole padded row in i (ni = idm+8 <= 1024 lanes, the E/W
// halo columns are ordinary columns of the row) and marches in j through a chunk of rows of one layer.  Every
// temporary lives in a ring of a few rows in LDS; sweep X works on row s - lag(X) while the first sweep works on row
// s, the lags being the stencil reaches in j.  Recomputation is limited to the 4-5 warm-up rows of a chunk.
// The stage's dependency graph has two chains that meet only in the final update, so there are two kernels:
//   k_mom_visc_march   "viscous chain": utotn/vtotn -> side-wall weights, uja.., dl2u/dl2v -> deformation ->
//                      viscosities -> longitudinal + lateral turbulent momentum fluxes; writes ONE plane per velocity
//                      point and layer, the flux divergence term of the update (:968-979, :1133-1143)
//                      visu = (uflux1(i,j)-uflux1(i-1,j)+uflux3-uflux2)/(scu2*max(dpu,onemm))
//   k_mom_cor_march    "Coriolis chain" + update: utotm/vtotm, uflux/vflux, dpmx -> vorticity, potential vorticity,
//                      kinetic energy -> Coriolis/advection term, stresses, pressure gradient average -> new u, v at
//                      both time levels, written to scratch planes (the in-place update of the reference would race
//                      with the neighbouring chunk's reads; k_mom_column_from reads the scratch planes and does the
//                      vertical pass into u, v, so no extra pass over the fields is needed).
// Expressions are those of stage_momtum.hip's kernels, operator for operator; loop bounds of every sweep as there.
// LDS: 37 rows (visc) / 21 or 35 rows (cor) of a strip.  Roofline: HBM; ~35 F moved by the two kernels.
#include "momtum_common.h"

// kk-level work-space slots of the fused path
enum { MF_VISU, MF_VISV, MF_UM, MF_UN, MF_VM, MF_VN, MF_NSLOT };
#define S2_DRAG 3      // 2-D work plane (written by k_mom_drag, stage_momtum.hip)

// Field pointers come out of the DevView in memory, so the compiler cannot know their address space and would emit
// flat loads -- which count on lgkmcnt as well as vmcnt, so that every LDS wait (and the wait in front of every
// barrier) also drains the global loads in flight.  Cast to the global address space: global_load, vmcnt only.
#ifdef BLOM_HOSTEMU
#define GLOBAL_AS
#else
#define GLOBAL_AS __attribute__((address_space(1)))
#endif
typedef const double GLOBAL_AS *gcd_t;
typedef double GLOBAL_AS *gd_t;
typedef const int GLOBAL_AS *gci_t;
#define GF(V, id) ((gcd_t)(V).f[id])

// ring of D rows of a strip: lane l of row r at row(r)[l]; two pad columns on either side so that l-1 .. l+2 of
// the edge lanes stay inside the row (what is read there belongs to no owned point)
template <int D> struct Ring {
  double *b;
  int nip;
  __device__ inline double *row(int r) const { return b + ((r + 16 * D) % D) * nip + 2; }
};

// chunk-major work item of this workgroup; XCD x (blockIdx % 8) walks a contiguous eighth of the items so that the
// workgroups of one XCD share the 2-D coefficient rows of their chunk in its L2
__device__ inline void march_item(int kk, int jj, int nchunk_, int nstrip, int &k, int &ja, int &jb, int &strip) {
  const bool chunk_major = nchunk_ < 0;
  const int nchunk = chunk_major ? -nchunk_ : nchunk_;
  const unsigned nitem = gridDim.x, lin = blockIdx.x;
  const unsigned xq = lin & 7u, sq = lin >> 3, q = nitem >> 3, rr = nitem & 7u;
  unsigned item = xq * q + (xq < rr ? xq : rr) + sq;
  strip = item % nstrip;          // the strips of a row chunk are neighbours in the numbering: they share rows in L2
  item /= nstrip;
  // chunk-major: the workgroups of an XCD work on the same rows of different layers at about the same time, so the
  // 2-D coefficient rows (25 planes, read by every layer) are served by its L2
  const int ch = chunk_major ? item / kk : item % nchunk;
  k = chunk_major ? item % kk : item / nchunk;
  const int rows = (jj + nchunk - 1) / nchunk;
  ja = 1 + ch * rows;
  jb = ja + rows - 1 < jj ? ja + rows - 1 : jj;
}

// ======================================================================================================
// viscous chain
// ======================================================================================================
// Every global load of a march step is issued at the top of the step, unconditionally and with clamped indices, before
// the first barrier: a workgroup of this kernel is alone on its CU (LDS), so nothing but the loads already in flight
// hides the memory latency of the six sweeps.  The first sweep's inputs are loaded one step ahead.
// mpack: bit 0 ip, 1 iu, 2 iv, 3 iq of a point in one word.
#define MP(m) ((m) & 1)
#define MU(m) (((m) >> 1) & 1)
#define MV(m) (((m) >> 2) & 1)
#define MQ(m) (((m) >> 3) & 1)
template <int BS>
__global__ __launch_bounds__(BS) void k_mom_visc_march(const DevView *__restrict__ Vp, int m, int n, int mm, int nn, int nchunk, int nstrip) {
  const DevView &V = *Vp;
  HIP_DYNAMIC_SHARED(double, lds)
  const int l = threadIdx.x, ni = V.ni, ii = V.ii, jj = V.jj, kk = V.kk;
  int k, ja, jb, strip;
  march_item(kk, jj, nchunk, nstrip, k, ja, jb, strip);
  if (ja > jb) return;
  // strip of the row: BS lanes, the outer HL / HR of them only feed their neighbours (stencil reach of the chain in i);
  // owned columns ox0..ox1 (x = i + 3), interior points i = 1..ii shared out over the strips
  constexpr int HL = 4, HR = 4, OW = BS - HL - HR;
  const int ox0 = NBDY + strip * OW, ox1 = (ox0 + OW - 1 < ii + NBDY - 1) ? ox0 + OW - 1 : ii + NBDY - 1;
  if (ox0 > ox1) return;
  const int x = ox0 - HL + l;
  const bool act = x < ni;
  const int i = x - (NBDY - 1);
  const bool own = x >= ox0 && x <= ox1;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np, okn = (size_t)(k + nn) * np;
  const size_t om = (size_t)(m - 1) * np, on = (size_t)(n - 1) * np;
  // scalars of the step in registers: left in the DevView they would be re-loaded inside the loop (the kernel stores
  // to memory the compiler cannot tell apart from it)
  const double tsfac = V.P.dlt / V.P.delt1;
  const double mdv2hi = V.P.mdv2hi, mdv2lo = V.P.mdv2lo, mdv4hi = V.P.mdv4hi, mdv4lo = V.P.mdv4lo;
  const double vsc2hi = V.P.vsc2hi, vsc2lo = V.P.vsc2lo, vsc4hi = V.P.vsc4hi, vsc4lo = V.P.vsc4lo;
  const gci_t mpk = (gci_t)V.m[I_mpack];
  // lanes outside every sweep's i-range (i < -1, i > ii+2) and rows outside the array read a neighbour's data instead,
  // which no sweep uses
  const int xl = x < 2 ? 2 : (x > ni - 3 ? ni - 3 : x);
  auto cidx = [&](int r) {
    const int rc = r < -2 ? -2 : (r > jj + 3 ? jj + 3 : r);
    return (size_t)xl + (size_t)ni * (rc + NBDY - 1);
  };

  double *lp = lds;
  constexpr int NP = BS + 4;
  auto take = [&](int d) { double *b = lp; lp += d * NP; return b; };
  const Ring<4> UTN{take(4), NP}, VTN{take(4), NP};
  const Ring<4> DL2U{take(4), NP};
  const Ring<3> VIB{take(3), NP}, DL2V{take(3), NP};
  const Ring<2> D1{take(2), NP}, D2{take(2), NP};
  const Ring<3> VS2U{take(3), NP}, VS4U{take(3), NP};
  const Ring<4> VS2V{take(4), NP}, VS4V{take(4), NP};
  const Ring<1> UFL1{take(1), NP};
  // fields that are only ever read at the lane that wrote them live in registers: value of row s-1 (written this
  // step by W), s-2, s-3 (read by U); ujb of row s-2 is read by V; vflux1 of rows s-3, s-4
  double wja1 = 0., wja2 = 0., wja3 = 0., wjb1 = 0., wjb2 = 0., wjb3 = 0., uja1 = 0., uja2 = 0., uja3 = 0.;
  double ujb1 = 0., ujb2 = 0., ujb3 = 0., wia1 = 0., wia2 = 0., wia3 = 0., wib1 = 0., wib2 = 0., wib3 = 0.;
  double via1 = 0., via2 = 0., via3 = 0., vfl3 = 0., vfl4 = 0.;

  const gcd_t f_u = GF(V, F_u) + okn, f_v = GF(V, F_v) + okn;
  const gcd_t f_ubf = GF(V, F_ubflxs_p) + on, f_vbf = GF(V, F_vbflxs_p) + on, f_pbun = GF(V, F_pbu) + on, f_pbvn = GF(V, F_pbv) + on;
  const gcd_t f_pbum = GF(V, F_pbu) + om, f_pbvm = GF(V, F_pbv) + om;
  const gcd_t f_pu0 = GF(V, F_pu) + ok, f_pu1 = GF(V, F_pu) + (size_t)(k + 1) * np;
  const gcd_t f_pv0 = GF(V, F_pv) + ok, f_pv1 = GF(V, F_pv) + (size_t)(k + 1) * np;
  const gcd_t f_dpu = GF(V, F_dpu) + okm, f_dpv = GF(V, F_dpv) + okm;
  const gcd_t scuy = GF(V, F_scuy), scvx = GF(V, F_scvx), scvy = GF(V, F_scvy), scux = GF(V, F_scux);
  const gcd_t scq2i = GF(V, F_scq2i), scp2i = GF(V, F_scp2i), difwgt = GF(V, F_difwgt);
  const gcd_t f_difmxp = GF(V, F_difmxp), f_difmxq = GF(V, F_difmxq), f_scpy = GF(V, F_scpy), f_scpx = GF(V, F_scpx);
  const gcd_t f_scqx = GF(V, F_scqx), f_scqy = GF(V, F_scqy), f_scu2 = GF(V, F_scu2), f_scv2 = GF(V, F_scv2);
  const gd_t o_visu = (gd_t)WK(V, MF_VISU) + ok, o_visv = (gd_t)WK(V, MF_VISV) + ok;
  const gd_t o_utotn = (gd_t)V.f[F_utotn], o_vtotn = (gd_t)V.f[F_vtotn];

  // inputs of the first sweep, one step ahead
  struct TIn { int mk; double u, ub, pbu, sy, v, vb, pbv, sx; };
  auto load_t = [&](int r) {
    const size_t c = cidx(r);
    TIn t;
    t.mk = mpk[c];
    t.u = f_u[c]; t.ub = f_ubf[c]; t.pbu = f_pbun[c]; t.sy = scuy[c];
    t.v = f_v[c]; t.vb = f_vbf[c]; t.pbv = f_pbvn[c]; t.sx = scvx[c];
    return t;
  };
  TIn tc = load_t(ja - 2);

  for (int s = ja - 2; s <= jb + 3; s++) {
    // ================= loads of this step =================
    const TIn tn = load_t(s + 1);
    // W, V (row s-1)
    const size_t cw = cidx(s - 1);
    const int w_m = mpk[cw], w_mw = mpk[cw - 1], w_ms = mpk[cw - ni];
    const double w_pu1 = f_pu1[cw], w_pu0 = f_pu0[cw], w_pbua = f_pbum[cw - ni], w_pbub = f_pbum[cw + ni];
    const double w_pv1 = f_pv1[cw], w_pv0 = f_pv0[cw], w_pbva = f_pbvm[cw - 1], w_pbvb = f_pbvm[cw + 1];
    const double v_scvy = scvy[cw], v_scvyw = scvy[cw - 1], v_scux = scux[cw], v_scuxs = scux[cw - ni], v_scq2i = scq2i[cw];
    const double v_scuy = scuy[cw], v_scuye = scuy[cw + 1], v_scvx = scvx[cw], v_scvxn = scvx[cw + ni], v_scp2i = scp2i[cw];
    const double sv_dw = difwgt[cw], sv_dws = difwgt[cw - ni];
    // S at u-points (row s-2)
    const size_t cs = cidx(s - 2);
    const int s_m = mpk[cs];
    const double su_dw = difwgt[cs], su_dww = difwgt[cs - 1];
    // F, U (row s-3)
    const size_t cf = cidx(s - 3);
    const int f_m = mpk[cf], f_me = mpk[cf + 1], f_me2 = mpk[cf + 2], f_mn = mpk[cf + ni], f_mn2 = mpk[cidx(s - 2) + ni];
    const int f_mw = mpk[cf - 1], f_ms = mpk[cf - ni];
    const double v_difmxp = f_difmxp[cf], v_scpy = f_scpy[cf], v_scpx = f_scpx[cf];
    const double dpu_c = f_dpu[cf], dpu_e = f_dpu[cf + 1], dpu_s = f_dpu[cf - ni], dpu_n = f_dpu[cf + ni];
    const double dpv_c = f_dpv[cf], dpv_n = f_dpv[cf + ni], dpv_w = f_dpv[cf - 1], dpv_e = f_dpv[cf + 1];
    const double dmq_c = f_difmxq[cf], dmq_n = f_difmxq[cf + ni], dmq_e = f_difmxq[cf + 1];
    const double scqx_c = f_scqx[cf], scqx_n = f_scqx[cf + ni], scqy_c = f_scqy[cf], scqy_e = f_scqy[cf + 1];
    const double scu2_c = f_scu2[cf], scv2_c = f_scv2[cf];

    // ---- T: total velocities at the old time level, row s (:408-431; rows -1..jj+2, i = -1..ii+2) ------------
    if (act && s >= -1 && s <= jj + 2 && i >= -1 && i <= ii + 2) {
      const size_t c = (size_t)x + (size_t)ni * (s + NBDY - 1);
      double un = 0., vn = 0.;
      if (MU(tc.mk)) {
        un = tc.u + tc.ub * tsfac / (tc.pbu * tc.sy);
        // the reference's module array utotn is left holding the last layer's values outside the interior
        if (k == kk - 1) o_utotn[c] = un;
      }
      if (MV(tc.mk)) {
        vn = tc.v + tc.vb * tsfac / (tc.pbv * tc.sx);
        if (k == kk - 1) o_vtotn[c] = vn;
      }
      UTN.row(s)[l] = un;
      VTN.row(s)[l] = vn;
    }
    __syncthreads();
    // ---- W: side-wall weights, auxiliary velocities, del2 fields, row s-1 (:438-472) ---------------------------
    {
      const int r = s - 1;
      if (act && r >= -1 && r <= jj + 2 && i >= 0 && i <= ii + 2) {
        double uja = 0., ujb = 0., d2u = 0.;
        if (MU(w_m)) {
          const double *utm = UTN.row(r - 1), *ut0 = UTN.row(r), *utp = UTN.row(r + 1);
          const double den = fmax2(w_pu1 - w_pu0, EPSILP);
          const double wa = fmax2(0., fmin2(1., (w_pu1 - w_pbua) / den));
          const double wb = fmax2(0., fmin2(1., (w_pu1 - w_pbub) / den));
          const double un = ut0[l];
          uja = (1. - wa) * utm[l] + wa * SLIP * un;
          ujb = (1. - wb) * utp[l] + wb * SLIP * un;
          d2u = un - .25 * (ut0[l + 1] + ut0[l - 1] + uja + ujb);
          wja1 = wa;
          wjb1 = wb;
        }
        uja1 = uja;
        ujb1 = ujb;
        DL2U.row(r)[l] = d2u;
      }
      if (act && r >= 0 && r <= jj + 2 && i >= -1 && i <= ii + 2) {
        double via = 0., vib = 0., d2v = 0.;
        if (MV(w_m)) {
          const double *vtm = VTN.row(r - 1), *vt0 = VTN.row(r), *vtp = VTN.row(r + 1);
          const double den = fmax2(w_pv1 - w_pv0, EPSILP);
          const double wa = fmax2(0., fmin2(1., (w_pv1 - w_pbva) / den));
          const double wb = fmax2(0., fmin2(1., (w_pv1 - w_pbvb) / den));
          const double vn = vt0[l];
          via = (1. - wa) * vt0[l - 1] + wa * SLIP * vn;
          vib = (1. - wb) * vt0[l + 1] + wb * SLIP * vn;
          d2v = vn - .25 * (vtp[l] + vtm[l] + via + vib);
          wia1 = wa;
          wib1 = wb;
        }
        via1 = via;
        VIB.row(r)[l] = vib;
        DL2V.row(r)[l] = d2v;
      }
    }
    __syncthreads();
    // ---- V: deformation, row s-1 (:500-507, :534-541, :549-559, :577-585) ---------------------------------------
    {
      const int r = s - 1;
      if (act && r >= 0 && r <= jj + 2 && i >= 0 && i <= ii + 2) {          // defor2 at q-points
        const double *ut0 = UTN.row(r), *utm = UTN.row(r - 1), *vt0 = VTN.row(r);
        bool have = false;
        double d2 = 0.;
        if (MV(w_m) && !MV(w_mw)) { const double t = vt0[l] * (1. - SLIP) * v_scvy; d2 = t * t * v_scq2i; have = true; }
        else if (MV(w_mw) && !MV(w_m)) { const double t = vt0[l - 1] * (1. - SLIP) * v_scvyw; d2 = t * t * v_scq2i; have = true; }
        if (MU(w_m) && !MU(w_ms)) { const double t = ut0[l] * (1. - SLIP) * v_scux; d2 = t * t * v_scq2i; have = true; }
        else if (MU(w_ms) && !MU(w_m)) { const double t = utm[l] * (1. - SLIP) * v_scuxs; d2 = t * t * v_scq2i; have = true; }
        if (MQ(w_m)) {
          const double t = VIB.row(r)[l - 1] * v_scvy - via1 * v_scvyw + ujb2 * v_scux - uja1 * v_scuxs;
          d2 = t * t * v_scq2i;
          have = true;
        }
        if (have) D2.row(r)[l] = d2;
      }
      if (act && r >= -1 && r <= jj + 1 && i >= -1 && i <= ii + 1 && MP(w_m)) {   // defor1 at p-points
        const double *ut0 = UTN.row(r), *vt0 = VTN.row(r), *vtp = VTN.row(r + 1);
        const double t = (ut0[l + 1] * v_scuye - ut0[l] * v_scuy) - (vtp[l] * v_scvxn - vt0[l] * v_scvx);
        D1.row(r)[l] = t * t * v_scp2i;
      }
    }
    __syncthreads();
    // ---- S: deformation dependent viscosities (:829-841 at u-points, row s-2; :988-1000 at v-points, row s-1) ----
    if (act && i >= 0 && i <= ii + 1) {
      {
        const int r = s - 2;
        if (r >= 0 && r <= jj + 1 && MU(s_m)) {
          const double *d1 = D1.row(r), *d2 = D2.row(r), *d2p = D2.row(r + 1);
          const double q = .5 * (su_dww + su_dw);
          const double deform = sqrt(.5 * (d1[l] + d1[l - 1] + d2[l] + d2p[l]));
          VS2U.row(r)[l] = fmax2(q * mdv2hi + (1. - q) * mdv2lo, (q * vsc2hi + (1. - q) * vsc2lo) * deform);
          VS4U.row(r)[l] = fmax2(q * mdv4hi + (1. - q) * mdv4lo, (q * vsc4hi + (1. - q) * vsc4lo) * deform);
        }
      }
      {
        const int r = s - 1;
        if (r >= 0 && r <= jj + 1 && MV(w_m)) {
          const double *d1 = D1.row(r), *d1m = D1.row(r - 1), *d2 = D2.row(r);
          const double q = .5 * (sv_dws + sv_dw);
          const double deform = sqrt(.5 * (d1[l] + d1m[l] + d2[l] + d2[l + 1]));
          VS2V.row(r)[l] = fmax2(q * mdv2hi + (1. - q) * mdv2lo, (q * vsc2hi + (1. - q) * vsc2lo) * deform);
          VS4V.row(r)[l] = fmax2(q * mdv4hi + (1. - q) * mdv4lo, (q * vsc4hi + (1. - q) * vsc4lo) * deform);
        }
      }
    }
    __syncthreads();
    // ---- F: longitudinal turbulent momentum fluxes at p-points, row s-3 (:860-873, :1019-1034) -----------------
    {
      const int r = s - 3;
      if (act && r >= 0 && r <= jj && i >= 0 && i <= ii && MP(f_m)) {
        if (r >= 1 && MU(f_m) + MU(f_me) > 0) {
          const double *ut0 = UTN.row(r), *dl2 = DL2U.row(r);
          const double *v2r = VS2U.row(r), *v4r = VS4U.row(r);
          const double dpxy = fmax2(dpu_c, ONEMM), dpib = fmax2(dpu_e, ONEMM);
          // viscosity extended one point beyond wet u-segments (:845-856), cf. ext_i
          const int m0 = MU(f_m), m1 = MU(f_me), m2 = MU(f_me2);
          const double v2 = (m0 ? v2r[l] : (m1 ? v2r[l + 1] : v2r[l - 1])) + (m1 ? v2r[l + 1] : (m2 ? v2r[l + 2] : v2r[l]));
          const double v4 = (m0 ? v4r[l] : (m1 ? v4r[l + 1] : v4r[l - 1])) + (m1 ? v4r[l + 1] : (m2 ? v4r[l + 2] : v4r[l]));
          UFL1.row(r)[l] = fmin2(v_difmxp, v2 * v_scpy) * hfharm(dpxy, dpib) * (ut0[l] - ut0[l + 1]) +
                           fmin2(.125 * v_difmxp, v4 * v_scpy) * hfharm(dpxy, dpib) * (dl2[l] - dl2[l + 1]);
        }
        if (i >= 1 && MV(f_m) + MV(f_mn) > 0) {
          const double *vt0 = VTN.row(r), *vtp = VTN.row(r + 1);
          const double *dl2 = DL2V.row(r), *dl2p = DL2V.row(r + 1);
          const double dpxy = fmax2(dpv_c, ONEMM), dpjb = fmax2(dpv_n, ONEMM);
          const int m0 = MV(f_m), m1 = MV(f_mn), m2 = MV(f_mn2);
          const double *a2m = VS2V.row(r - 1), *a20 = VS2V.row(r), *a2p = VS2V.row(r + 1), *a2q = VS2V.row(r + 2);
          const double *a4m = VS4V.row(r - 1), *a40 = VS4V.row(r), *a4p = VS4V.row(r + 1), *a4q = VS4V.row(r + 2);
          const double v2 = (m0 ? a20[l] : (m1 ? a2p[l] : a2m[l])) + (m1 ? a2p[l] : (m2 ? a2q[l] : a20[l]));
          const double v4 = (m0 ? a40[l] : (m1 ? a4p[l] : a4m[l])) + (m1 ? a4p[l] : (m2 ? a4q[l] : a40[l]));
          vfl3 = fmin2(v_difmxp, v2 * v_scpx) * hfharm(dpxy, dpjb) * (vt0[l] - vtp[l]) +
                           fmin2(.125 * v_difmxp, v4 * v_scpx) * hfharm(dpxy, dpjb) * (dl2[l] - dl2p[l]);
        }
      }
    }
    __syncthreads();
    // ---- U: lateral turbulent momentum fluxes and the flux divergence term, row s-3 (:879-913, :1040-1076) -----
    {
      const int r = s - 3;
      const size_t c = (size_t)x + (size_t)ni * (r + NBDY - 1);
      if (act && own && r >= ja && r <= jb && i >= 1 && i <= ii) {
        if (MU(f_m)) {
          const double wja = wja3, wjb = wjb3;
          const double dpxy = fmax2(dpu_c, ONEMM);
          double dpja = fmax2(dpu_s, ONEMM);
          dpja = dpja + wja * (dpxy - dpja);
          double dpjb = fmax2(dpu_n, ONEMM);
          dpjb = dpjb + wjb * (dpxy - dpjb);
          const double v2c = VS2U.row(r)[l], v4c = VS4U.row(r)[l];
          const double vsc2a = MU(f_ms) == 0 ? v2c : VS2U.row(r - 1)[l], vsc4a = MU(f_ms) == 0 ? v4c : VS4U.row(r - 1)[l];
          const double vsc2b = MU(f_mn) == 0 ? v2c : VS2U.row(r + 1)[l], vsc4b = MU(f_mn) == 0 ? v4c : VS4U.row(r + 1)[l];
          const double un = UTN.row(r)[l], d2 = DL2U.row(r)[l];
          const double dl2uja = (1. - wja) * DL2U.row(r - 1)[l] + wja * SLIP * d2;          // :594-597
          const double dl2ujb = (1. - wjb) * DL2U.row(r + 1)[l] + wjb * SLIP * d2;
          const double uflux2 = fmin2(dmq_c, (v2c + vsc2a) * scqx_c) * hfharm(dpja, dpxy) * (uja3 - un) +
                                fmin2(.125 * dmq_c, (v4c + vsc4a) * scqx_c) * hfharm(dpja, dpxy) * (dl2uja - d2);
          const double uflux3 = fmin2(dmq_n, (v2c + vsc2b) * scqx_n) * hfharm(dpjb, dpxy) * (un - ujb3) +
                                fmin2(.125 * dmq_n, (v4c + vsc4b) * scqx_n) * hfharm(dpjb, dpxy) * (d2 - dl2ujb);
          const double *uflux1 = UFL1.row(r);
          o_visu[c] = (uflux1[l] - uflux1[l - 1] + uflux3 - uflux2) / (scu2_c * fmax2(dpu_c, ONEMM));
        }
        if (MV(f_m)) {
          const double wia = wia3, wib = wib3;
          const double dpxy = fmax2(dpv_c, ONEMM);
          double dpia = fmax2(dpv_w, ONEMM);
          dpia = dpia + wia * (dpxy - dpia);
          double dpib = fmax2(dpv_e, ONEMM);
          dpib = dpib + wib * (dpxy - dpib);
          const double *vsc2 = VS2V.row(r), *vsc4 = VS4V.row(r), *dl2v = DL2V.row(r);
          const double vsc2a = MV(f_mw) == 0 ? vsc2[l] : vsc2[l - 1], vsc4a = MV(f_mw) == 0 ? vsc4[l] : vsc4[l - 1];
          const double vsc2b = MV(f_me) == 0 ? vsc2[l] : vsc2[l + 1], vsc4b = MV(f_me) == 0 ? vsc4[l] : vsc4[l + 1];
          const double vn = VTN.row(r)[l], d2 = dl2v[l];
          const double dl2via = (1. - wia) * dl2v[l - 1] + wia * SLIP * d2;          // :602-605
          const double dl2vib = (1. - wib) * dl2v[l + 1] + wib * SLIP * d2;
          const double vflux2 = fmin2(dmq_c, (vsc2[l] + vsc2a) * scqy_c) * hfharm(dpia, dpxy) * (via3 - vn) +
                                fmin2(.125 * dmq_c, (vsc4[l] + vsc4a) * scqy_c) * hfharm(dpia, dpxy) * (dl2via - d2);
          const double vflux3 = fmin2(dmq_e, (vsc2[l] + vsc2b) * scqy_e) * hfharm(dpib, dpxy) * (vn - VIB.row(r)[l]) +
                                fmin2(.125 * dmq_e, (vsc4[l] + vsc4b) * scqy_e) * hfharm(dpib, dpxy) * (d2 - dl2vib);
          o_visv[c] = (vfl3 - vfl4 + vflux3 - vflux2) / (scv2_c * fmax2(dpv_c, ONEMM));
        }
      }
    }
    __syncthreads();
    wja3 = wja2; wja2 = wja1; wjb3 = wjb2; wjb2 = wjb1; uja3 = uja2; uja2 = uja1; ujb3 = ujb2; ujb2 = ujb1;
    wia3 = wia2; wia2 = wia1; wib3 = wib2; wib2 = wib1; via3 = via2; via2 = via1; vfl4 = vfl3;
    tc = tn;
  }
}

// ======================================================================================================
// Coriolis chain + update
// ======================================================================================================
template <int BS, bool ENEDIS>
__global__ __launch_bounds__(BS) void k_mom_cor_march(const DevView *__restrict__ Vp, int m, int n, int mm, int nn, int nchunk, int nstrip) {
  const DevView &V = *Vp;
  HIP_DYNAMIC_SHARED(double, lds)
  const int l = threadIdx.x, ni = V.ni, ii = V.ii, jj = V.jj, kk = V.kk;
  int k, ja, jb, strip;
  march_item(kk, jj, nchunk, nstrip, k, ja, jb, strip);
  if (ja > jb) return;
  // strip of the row: BS lanes, the outer HL / HR of them only feed their neighbours (stencil reach of the chain in i);
  // owned columns ox0..ox1 (x = i + 3), interior points i = 1..ii shared out over the strips
  constexpr int HL = 2, HR = 2, OW = BS - HL - HR;
  const int ox0 = NBDY + strip * OW, ox1 = (ox0 + OW - 1 < ii + NBDY - 1) ? ox0 + OW - 1 : ii + NBDY - 1;
  if (ox0 > ox1) return;
  const int x = ox0 - HL + l;
  const bool act = x < ni;
  const int i = x - (NBDY - 1);
  const bool own = x >= ox0 && x <= ox1;
  const size_t np = V.nplane, ok = (size_t)k * np, okm = (size_t)(k + mm) * np, okn = (size_t)(k + nn) * np;
  const size_t om = (size_t)(m - 1) * np, on = (size_t)(n - 1) * np;
  const double delt1 = V.P.delt1, tsfac = V.P.dlt / V.P.delt1, cutoff = ONEM, thkbop = THKBOT * ONEM;
  const double wuv1 = V.P.wuv1, wuv2 = V.P.wuv2;
  const int mommth = V.P.mommth;
  const bool last_chunk = jb == jj;
  const gci_t mpk = (gci_t)V.m[I_mpack];
  const int xl = x < 2 ? 2 : (x > ni - 3 ? ni - 3 : x);
  auto cidx = [&](int r) {
    const int rc = r < -2 ? -2 : (r > jj + 3 ? jj + 3 : r);
    return (size_t)xl + (size_t)ni * (rc + NBDY - 1);
  };

  double *lp = lds;
  constexpr int NP = BS + 4;
  auto take = [&](int d) { double *b = lp; lp += d * NP; return b; };
  const Ring<3> UTM{take(3), NP}, VTM{take(3), NP};
  const Ring<4> UFX{take(4), NP};
  const Ring<3> VFX{take(3), NP}, DPMX{take(3), NP};
  const Ring<2> PV{take(2), NP};
  const Ring<3> KE{take(3), NP};
  const Ring<4> UHMN{take(ENEDIS ? 4 : 0), NP}, UHMX{take(ENEDIS ? 4 : 0), NP};
  const Ring<3> VHMN{take(ENEDIS ? 3 : 0), NP}, VHMX{take(ENEDIS ? 3 : 0), NP};

  const gcd_t dp = GF(V, F_dp) + okm, f_um = GF(V, F_u) + okm, f_vm = GF(V, F_v) + okm, f_un = GF(V, F_u) + okn, f_vn = GF(V, F_v) + okn;
  const gcd_t f_ubfm = GF(V, F_ubflxs_p) + om, f_vbfm = GF(V, F_vbflxs_p) + om, f_pbum = GF(V, F_pbu) + om, f_pbvm = GF(V, F_pbv) + om;
  const gcd_t f_ubfn = GF(V, F_ubflxs_p) + on, f_vbfn = GF(V, F_vbflxs_p) + on, f_pbun = GF(V, F_pbu) + on, f_pbvn = GF(V, F_pbv) + on;
  const gcd_t f_dpu = GF(V, F_dpu) + okm, f_dpv = GF(V, F_dpv) + okm;
  const gcd_t scuy = GF(V, F_scuy), scvx = GF(V, F_scvx), scvy = GF(V, F_scvy), scux = GF(V, F_scux), scq2i = GF(V, F_scq2i);
  const gcd_t scu2 = GF(V, F_scu2), scv2 = GF(V, F_scv2), scp2 = GF(V, F_scp2), corioq = GF(V, F_corioq);
  const gcd_t drag = (gcd_t)WK2(V, S2_DRAG), p0 = GF(V, F_p) + ok, p1 = GF(V, F_p) + (size_t)(k + 1) * np, p_1 = GF(V, F_p) + np;
  const gcd_t pgfx_m = GF(V, F_pgfx) + okm, pgfx_n = GF(V, F_pgfx) + okn, pgfx_o = GF(V, F_pgfx_o) + ok;
  const gcd_t pgfy_m = GF(V, F_pgfy) + okm, pgfy_n = GF(V, F_pgfy) + okn, pgfy_o = GF(V, F_pgfy_o) + ok;
  const gcd_t dpuold = GF(V, F_dpuold) + ok, dpvold = GF(V, F_dpvold) + ok, ubcors = GF(V, F_ubcors_p), vbcors = GF(V, F_vbcors_p);
  const gcd_t scuxi = GF(V, F_scuxi), scvyi = GF(V, F_scvyi), taux = GF(V, F_taux), tauy = GF(V, F_tauy);
  const bool hybrid = V.P.vcoord_tag != 1;
  const gcd_t mu_nl = GF(V, F_mu_nonloc), mv_nl = GF(V, F_mv_nonloc);
  const gcd_t visu = (gcd_t)WK(V, MF_VISU) + ok, visv = (gcd_t)WK(V, MF_VISV) + ok;
  const gd_t o_um = (gd_t)WK(V, MF_UM) + ok, o_un = (gd_t)WK(V, MF_UN) + ok, o_vm = (gd_t)WK(V, MF_VM) + ok, o_vn = (gd_t)WK(V, MF_VN) + ok;
  const gd_t o_absvor = (gd_t)V.f[F_absvor] + ok, o_dpvor = (gd_t)V.f[F_dpvor] + ok;

  // Inputs of the first sweep are loaded one step ahead; what the later sweeps need of the same rows (the wet masks
  // and dp of rows s-1, s-2 for the vorticity, u, v, dpu, dpv and the 2-D coefficients of row s-2 for the update)
  // is carried in registers from step to step instead of being read again: the second read of a row, two march
  // steps later and with ~2000 waves streaming through the same L2, mostly missed it.
  struct TIn { int mk, mkw; double dc, dw, u, ub, pbu, sy, dpu, v, vb, pbv, sx, dpv; };
  auto load_t = [&](int r) {
    const size_t c = cidx(r);
    TIn t;
    t.mk = mpk[c]; t.mkw = mpk[c - 1];
    t.dc = dp[c]; t.dw = dp[c - 1];
    t.u = f_um[c]; t.ub = f_ubfm[c]; t.pbu = f_pbum[c]; t.sy = scuy[c]; t.dpu = f_dpu[c];
    t.v = f_vm[c]; t.vb = f_vbfm[c]; t.pbv = f_pbvm[c]; t.sx = scvx[c]; t.dpv = f_dpv[c];
    return t;
  };
  TIn tpp = load_t(ja - 3), tp = load_t(ja - 2), tc = load_t(ja - 1);      // rows s-2, s-1, s
  double pr_p0 = p0[cidx(ja - 4)], pr_p1 = p1[cidx(ja - 4)], pr_drag = drag[cidx(ja - 4)];   // row s-3

  for (int s = ja - 1; s <= jb + 2; s++) {
    // ================= loads of this step =================
    const TIn tn = load_t(s + 1);
    // V (row s-1): 2-D coefficients
    const size_t cv = cidx(s - 1);
    const int v_m = tp.mk, v_mw = tp.mkw, v_ms = tpp.mk;
    const double v_scvy = scvy[cv], v_scvyw = scvy[cv - 1], v_scux = scux[cv], v_scuxs = scux[cv - ni], v_scq2i = scq2i[cv];
    const double v_dc = tp.dc, v_dw = tp.dw, v_ds = tpp.dc, v_dsw = tpp.dw, v_cor = corioq[cv];
    const double v_scu2 = scu2[cv], v_scu2e = scu2[cv + 1], v_scv2 = scv2[cv], v_scv2n = scv2[cv + ni], v_scp2 = scp2[cv];
    // U (row s-2)
    const size_t cu = cidx(s - 2);
    const int u_m = tpp.mk;
    const double u_drag = drag[cu], u_dragw = drag[cu - 1], u_drags = pr_drag;
    const double u_p0 = p0[cu], u_p0w = p0[cu - 1], u_p0s = pr_p0, u_p1 = p1[cu], u_p1w = p1[cu - 1], u_p1s = pr_p1;
    const double u_dpu = tpp.dpu, u_pbum = tpp.pbu, u_ukm = tpp.u, u_ukn = f_un[cu], u_ubfn = f_ubfn[cu], u_pbun = f_pbun[cu];
    const double u_scuy = tpp.sy, u_pgm = pgfx_m[cu], u_pgo = pgfx_o[cu], u_pgn = pgfx_n[cu], u_dpuold = dpuold[cu];
    const double u_ubcors = ubcors[cu], u_scuxi = scuxi[cu], u_visu = visu[cu];
    const double u_dpv = tpp.dpv, u_pbvm = tpp.pbv, u_vkm = tpp.v, u_vkn = f_vn[cu], u_vbfn = f_vbfn[cu], u_pbvn = f_pbvn[cu];
    const double u_scvx = tpp.sx, u_pgym = pgfy_m[cu], u_pgyo = pgfy_o[cu], u_pgyn = pgfy_n[cu], u_dpvold = dpvold[cu];
    const double u_vbcors = vbcors[cu], u_scvyi = scvyi[cu], u_visv = visv[cu];
    // the first sweep's neighbours to the south come from the row loaded a step earlier
    const int t_mks = tp.mk;
    const double t_ds = tp.dc, t_dsw = tp.dw;

    // ---- T: total velocities at the mid time level, fluxes, dpmx, row s (:360-406; rows 0..jj+1 / dpmx 0..jj+2) ----
    if (act && s >= 0 && s <= jj + 2 && i >= 0 && i <= ii + 2) {
      double d = 8. * cutoff;
      if (MU(tc.mk)) d = fmax2(d, tc.dc + tc.dw);
      if (MU(t_mks)) d = fmax2(d, t_ds + t_dsw);
      if (MV(tc.mk)) d = fmax2(d, tc.dc + t_ds);
      if (MV(tc.mkw)) d = fmax2(d, tc.dw + t_dsw);
      DPMX.row(s)[l] = d;
      if (s <= jj + 1 && i <= ii + 1) {
        double ut = 0., uf = 0., vt = 0., vf = 0.;
        if (MU(tc.mk)) {
          ut = tc.u + tc.ub * tsfac / (tc.pbu * tc.sy);
          uf = ut * fmax2(tc.dpu, cutoff);
        }
        if (MV(tc.mk)) {
          vt = tc.v + tc.vb * tsfac / (tc.pbv * tc.sx);
          vf = vt * fmax2(tc.dpv, cutoff);
        }
        UTM.row(s)[l] = ut;
        UFX.row(s)[l] = uf;
        VTM.row(s)[l] = vt;
        VFX.row(s)[l] = vf;
        if (ENEDIS) {                                    // :662-715, rows 0..jj+1
          double a = 0., b = 0.;
          if (MU(tc.mk)) enedis_minmax(.5 * ut * (tc.dc + tc.dw), uf, a, b);
          UHMN.row(s)[l] = a; UHMX.row(s)[l] = b;
          a = 0.; b = 0.;
          if (MV(tc.mk)) enedis_minmax(.5 * vt * (tc.dc + t_ds), vf, a, b);
          VHMN.row(s)[l] = a; VHMX.row(s)[l] = b;
        }
      }
    }
    __syncthreads();
    // ---- V: vorticity / potential vorticity at q-points (:477-575) and kinetic energy (:613-629), row s-1 -------
    {
      const int r = s - 1;
      const size_t c = (size_t)x + (size_t)ni * (r + NBDY - 1);
      if (act && r >= 1 && r <= jj + 1 && i >= 1 && i <= ii + 1) {
        const double *utm0 = UTM.row(r), *utmm = UTM.row(r - 1), *vtm0 = VTM.row(r);
        const double *dx0 = DPMX.row(r), *dxm = DPMX.row(r - 1), *dxp = DPMX.row(r + 1);
        bool have = false;
        double vort = 0., dpv = 1.;
        if (MV(v_m) && !MV(v_mw)) {                      // first point of a v-segment, :479-486
          vort = vtm0[l] * (1. - SLIP) * v_scvy * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_dc + v_ds), dx0[l]), dx0[l + 1]);
          have = true;
        } else if (MV(v_mw) && !MV(v_m)) {               // one past the last point of a v-segment, :487-494
          vort = -vtm0[l - 1] * (1. - SLIP) * v_scvyw * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_dw + v_dsw), dx0[l - 1]), dx0[l]);
          have = true;
        }
        if (MU(v_m) && !MU(v_ms)) {                      // first point (in j) of a u-segment, :513-520
          vort = -utm0[l] * (1. - SLIP) * v_scux * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_dc + v_dw), dx0[l]), dxp[l]);
          have = true;
        } else if (MU(v_ms) && !MU(v_m)) {               // one past the last point, :521-528
          vort = utmm[l] * (1. - SLIP) * v_scuxs * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_ds + v_dsw), dxm[l]), dx0[l]);
          have = true;
        }
        if (MQ(v_m)) {                                   // interior (incl. promontories), :561-575
          vort = (vtm0[l] * v_scvy - vtm0[l - 1] * v_scvyw - utm0[l] * v_scux + utmm[l] * v_scuxs) * v_scq2i;
          double d = fmax2(2. * (v_dc + v_dw + v_ds + v_dsw), dx0[l]);
          d = fmax2(d, dx0[l - 1]);
          d = fmax2(d, dx0[l + 1]);
          d = fmax2(d, dxm[l]);
          d = fmax2(d, dxp[l]);
          dpv = .125 * d;
          have = true;
        }
        if (have) {
          const double av = vort + v_cor;
          if ((own || (x == ii + NBDY && ox1 == ii + NBDY - 1)) && ((r >= ja && r <= jb) || (last_chunk && r == jj + 1))) {
            o_absvor[c] = av;
            o_dpvor[c] = dpv;
          }
          PV.row(r)[l] = av / dpv;
        }
      }
      if (act && r >= 0 && r <= jj && i >= 0 && i <= ii && MP(v_m)) {
        const double *utm0 = UTM.row(r), *vtm0 = VTM.row(r), *vtmp = VTM.row(r + 1);
        const double ue = utm0[l + 1], uw = utm0[l], vn = vtmp[l], vs = vtm0[l];
        KE.row(r)[l] = .25 * (v_scu2 * (uw * uw) + v_scu2e * (ue * ue) + v_scv2 * (vs * vs) + v_scv2n * (vn * vn)) / v_scp2;
      }
    }
    __syncthreads();
    // ---- U: Coriolis/advection, stresses, pressure gradient, update of both time levels, row s-2 ----------------
    {
      const int r = s - 2;
      const size_t c = (size_t)x + (size_t)ni * (r + NBDY - 1);
      if (act && own && r >= ja && r <= jb && i >= 1 && i <= ii) {
        const double *pv0 = PV.row(r), *pvp = PV.row(r + 1), *ke0 = KE.row(r), *kem = KE.row(r - 1);
        if (MU(u_m)) {
          const double *vf0 = VFX.row(r), *vfp = VFX.row(r + 1);
          double cau;
          if (ENEDIS) {                                                            // enedis, :771-790
            const double *mx0 = VHMX.row(r), *mxp = VHMX.row(r + 1), *mn0 = VHMN.row(r), *mnp = VHMN.row(r + 1);
            const double utm = UTM.row(r)[l];
            double t1, t2;
            const double pn = pvp[l], pc = pv0[l];
            if (pn * utm == 0.) t1 = pn * ((mxp[l] + mxp[l - 1]) + (mnp[l] + mnp[l - 1])) * .5;
            else if (pn * utm < 0.) t1 = pn * (mxp[l] + mxp[l - 1]);
            else t1 = pn * (mnp[l] + mnp[l - 1]);
            if (pc * utm == 0.) t2 = pc * ((mx0[l] + mx0[l - 1]) + (mn0[l] + mn0[l - 1])) * .5;
            else if (pc * utm < 0.) t2 = pc * (mx0[l] + mx0[l - 1]);
            else t2 = pc * (mn0[l] + mn0[l - 1]);
            cau = .25 * (t1 + t2);
          } else if (mommth == 0)
            cau = .125 * (vf0[l] + vfp[l] + vf0[l - 1] + vfp[l - 1]) * (pv0[l] + pvp[l]);
          else
            cau = .25 * ((vf0[l] + vf0[l - 1]) * pv0[l] + (vfp[l] + vfp[l - 1]) * pvp[l]);
          // wind stress (isopyc_bulkml: top layer only), :919-936
          double stress = 0.;
          if (hybrid)                    // the other vertical coordinates: the stress spread by the non-local fractions, :937-946
            stress = -(mu_nl[c + (size_t)k * np] - mu_nl[c + (size_t)(k + 1) * np]) * taux[c] * GRAV * scux[c] / fmax2(ONEMM, u_dpu);
          else if (k == 0) stress = -2. * taux[c] * GRAV * scux[c] / (p_1[c] + p_1[c - 1]);
          const double pbu = u_pbum;
          const double ptopl = .5 * (fmin2(pbu, u_p0) + fmin2(pbu, u_p0w));
          const double pbotl = .5 * (fmin2(pbu, u_p1) + fmin2(pbu, u_p1w));
          const double q = .5 * (u_drag + u_dragw) * (fmax2(pbu - thkbop, pbotl) - fmax2(pbu - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                           fmax2(u_dpu, ONEMM);
          const double ukm = u_ukm, ukn = u_ukn;
          const double un = ukn + u_ubfn * tsfac / (u_pbun * u_scuy);               // utotn, :408-414
          const double botstr = -un * q / (1. + delt1 * q);
          const double pgf = (1. - 2. * WPGF) * u_pgm + WPGF * (u_pgo + u_pgn);
          o_um[c] = ukm * (wuv1 * u_dpu + ONEMM) + ukn * wuv2 * u_dpuold;
          const double ubrhs = u_ubcors * tsfac;                                    // :302
          o_un[c] = ukn + delt1 * (-u_scuxi * (-pgf + stress + (ke0[l] - ke0[l - 1])) + cau - ubrhs + botstr - u_visu);
        }
        if (MV(u_m)) {
          const double *uf0 = UFX.row(r), *ufm = UFX.row(r - 1);
          double cav;
          if (ENEDIS) {                                                            // enedis, :793-812
            const double *mx0 = UHMX.row(r), *mxm = UHMX.row(r - 1), *mn0 = UHMN.row(r), *mnm = UHMN.row(r - 1);
            const double vtm = VTM.row(r)[l];
            double t1, t2;
            const double pe = pv0[l + 1], pc = pv0[l];
            if (pe * vtm == 0.) t1 = pe * ((mx0[l + 1] + mxm[l + 1]) + (mn0[l + 1] + mnm[l + 1])) * .5;
            else if (pe * vtm > 0.) t1 = pe * (mx0[l + 1] + mxm[l + 1]);
            else t1 = pe * (mn0[l + 1] + mnm[l + 1]);
            if (pc * vtm == 0.) t2 = pc * ((mx0[l] + mxm[l]) + (mn0[l] + mnm[l])) * .5;
            else if (pc * vtm > 0.) t2 = pc * (mx0[l] + mxm[l]);
            else t2 = pc * (mn0[l] + mnm[l]);
            cav = -.25 * (t1 + t2);
          } else if (mommth == 0)
            cav = -.125 * (uf0[l] + uf0[l + 1] + ufm[l] + ufm[l + 1]) * (pv0[l] + pv0[l + 1]);
          else
            cav = -.25 * ((uf0[l] + ufm[l]) * pv0[l] + (uf0[l + 1] + ufm[l + 1]) * pv0[l + 1]);
          double stress = 0.;
          if (hybrid)                    // :1100-1109
            stress = -(mv_nl[c + (size_t)k * np] - mv_nl[c + (size_t)(k + 1) * np]) * tauy[c] * GRAV * scvy[c] / fmax2(ONEMM, u_dpv);
          else if (k == 0) stress = -2. * tauy[c] * GRAV * scvy[c] / (p_1[c] + p_1[c - ni]);
          const double pbv = u_pbvm;
          const double ptopl = .5 * (fmin2(pbv, u_p0) + fmin2(pbv, u_p0s));
          const double pbotl = .5 * (fmin2(pbv, u_p1) + fmin2(pbv, u_p1s));
          const double q = .5 * (u_drag + u_drags) * (fmax2(pbv - thkbop, pbotl) - fmax2(pbv - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                           fmax2(u_dpv, ONEMM);
          const double vkm = u_vkm, vkn = u_vkn;
          const double vn = vkn + u_vbfn * tsfac / (u_pbvn * u_scvx);               // vtotn, :424-430
          const double botstr = -vn * q / (1. + delt1 * q);
          const double pgf = (1. - 2. * WPGF) * u_pgym + WPGF * (u_pgyo + u_pgyn);
          o_vm[c] = vkm * (wuv1 * u_dpv + ONEMM) + vkn * wuv2 * u_dpvold;
          const double vbrhs = u_vbcors * tsfac;                                    // :307
          o_vn[c] = vkn + delt1 * (-u_scvyi * (-pgf + stress + (ke0[l] - kem[l])) + cav - vbrhs + botstr - u_visv);
        }
      }
    }
    __syncthreads();
    pr_p0 = u_p0; pr_p1 = u_p1; pr_drag = u_drag;
    tpp = tp; tp = tc; tc = tn;
  }
}

// ---- :1154-1267 vertical pass, reading the updated velocities from the scratch planes ----------------------------
// (k_mom_column of stage_momtum.hip with u(km), u(kn) taken from MF_UM/MF_UN, v likewise)
__global__ __launch_bounds__(64) void k_mom_column_from(const DevView *__restrict__ Vp, int m, int mm, int nn) {
  const DevView &V = *Vp;
  unsigned bx_, by_;
  xcd_block(bx_, by_);
  const int t_ = bx_ * blockDim.x + threadIdx.x;
  if (t_ >= V.nplane) return;
  const int i = t_ % V.ni - (NBDY - 1), j = t_ / V.ni - (NBDY - 1);
  const size_t c = t_;
  if (j < 1 || j > V.jj || i < 1 || i > V.ii) return;
  const bool isv = by_ == 1;
  if (!(isv ? V.m[I_iv][c] : V.m[I_iu][c])) return;
  const size_t np = V.nplane;
  const int kk = V.kk;
  double *u = isv ? V.f[F_v] : V.f[F_u];
  const double *sm = WK(V, isv ? MF_VM : MF_UM), *sn = WK(V, isv ? MF_VN : MF_UN);
  const double *dpu = isv ? V.f[F_dpv] : V.f[F_dpu], *dpuold = isv ? V.f[F_dpvold] : V.f[F_dpuold];
  const double umax = (isv ? V.f[F_vmax] : V.f[F_umax])[c];
  const double ub = (isv ? V.f[F_vb] : V.f[F_ub])[c + (size_t)(m - 1) * np];
  const double wuv1 = V.P.wuv1, wuv2 = V.P.wuv2;
  double tot = 0., uabove = 0.;
  const double *dpum = dpu + c + (size_t)mm * np, *dpun = dpu + c + (size_t)nn * np;
  double *um = u + c + (size_t)mm * np, *un_ = u + c + (size_t)nn * np;
  sm += c; sn += c; dpuold += c;
  for (int k0 = 0; k0 < kk; k0 += COLUMN_U) {                // COLUMN_U levels' loads in flight (blomgpu_internal.h)
    double a[COLUMN_U], b[COLUMN_U], d[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const size_t o = (size_t)(k0 + u < kk ? k0 + u : kk - 1) * np;
      a[u] = dpum[o]; b[u] = dpun[o]; d[u] = sn[o];
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 + u;
      if (k < kk) {
        const double dn = b[u];
        const double q = fmin2(fmin2(a[u], dn), ONEM);
        double un = d[u];
        const double ukan = k == 0 ? un : uabove;                              // kan = max(1,k-1)+nn
        un = (un * q + ukan * (ONEM - q)) / ONEM;
        un = fmax2(-umax, fmin2(umax, un + ub)) - ub;
        un_[(size_t)k * np] = un;
        uabove = un;
        tot = tot + un * dn;
      }
    }
  }
  tot = tot / (isv ? V.f[F_pbv_p] : V.f[F_pbu_p])[c];
  double pacc = (isv ? V.f[F_pv] : V.f[F_pu])[c];
  double *pun = (isv ? V.f[F_pv] : V.f[F_pu]) + c;
  for (int k0 = 0; k0 < kk; k0 += COLUMN_U) {
    double a[COLUMN_U], b[COLUMN_U], d[COLUMN_U], e[COLUMN_U], f[COLUMN_U];
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const size_t o = (size_t)(k0 + u < kk ? k0 + u : kk - 1) * np;
      a[u] = dpum[o]; b[u] = dpun[o]; d[u] = un_[o]; e[u] = sm[o]; f[u] = dpuold[o];
    }
#pragma unroll
    for (int u = 0; u < COLUMN_U; u++) {
      const int k = k0 + u;
      if (k < kk) {
        const double dn = b[u];
        const double un = d[u] - tot;
        un_[(size_t)k * np] = un;
        um[(size_t)k * np] = (e[u] + un * wuv2 * dn) / (wuv1 * a[u] + ONEMM + wuv2 * (f[u] + dn));
        pacc = pacc + dn;
        pun[(size_t)(k + 1) * np] = pacc;
      }
    }
  }
  (isv ? V.f[F_vtotn] : V.f[F_utotn])[c] = tot * (1. / V.P.delt1);
}

template <int BS>
static void launch_marches(blomgpu_ctx *c, int m, int n, int mm, int nn, int nca, int ncb) {
  const DevView &h = c->h;
  const int nsa = (h.ii + (BS - 8) - 1) / (BS - 8), nsb = (h.ii + (BS - 4) - 1) / (BS - 4);
  const size_t la = sizeof(double) * 37 * (BS + 4), lb = sizeof(double) * (h.P.mommth == 2 ? 35 : 21) * (BS + 4);
  // more than 64 KB of dynamic LDS has to be asked for
  (void)hipFuncSetAttribute((const void *)k_mom_visc_march<BS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)la);
  (void)hipFuncSetAttribute((const void *)k_mom_cor_march<BS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
  (void)hipFuncSetAttribute((const void *)k_mom_cor_march<BS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
  hipLaunchKernelGGL(k_mom_visc_march<BS>, dim3(h.kk * nca * nsa), dim3(BS), la, c->stream, c->d, m, n, mm, nn, c->momtum_order ? nca : -nca, nsa);
  if (h.P.mommth == 2)
    hipLaunchKernelGGL((k_mom_cor_march<BS, true>), dim3(h.kk * ncb * nsb), dim3(BS), lb, c->stream, c->d, m, n, mm, nn, c->momtum_order ? ncb : -ncb, nsb);
  else
    hipLaunchKernelGGL((k_mom_cor_march<BS, false>), dim3(h.kk * ncb * nsb), dim3(BS), lb, c->stream, c->d, m, n, mm, nn, c->momtum_order ? ncb : -ncb, nsb);
}

// the layer loop and the vertical pass of momtum; the caller (st_momtum) has done p/pu/pv, the drag and difwgt's halo
int st_momtum_fused_layers(blomgpu_ctx *c, int m, int n, int mm, int nn) {
  const DevView &h = c->h;
  if (h.nwk < MF_NSLOT) return ctx_fail(c, "momtum: device work space too small");
  // One wavefront per workgroup (the barriers of the march then cost nothing and 5 / 14 workgroups fit a CU by their
  // LDS): strips of 56 / 60 owned columns.  Chunks in j: about one round of workgroups on the chip.
  const int cus = c->num_cus > 0 ? c->num_cus : 256;
  const int bs = c->momtum_bs > 0 ? c->momtum_bs : 64;
  auto chunks = [&](int slots, int nstrip, int opt) {
    int nc = opt > 0 ? opt : slots / (h.kk * nstrip);
    if (nc > (h.jj + 15) / 16) nc = (h.jj + 15) / 16;      // at least 16 rows per chunk: the warm-up is 4-5 rows
    return nc < 1 ? 1 : nc;
  };
  const int nsa = (h.ii + (bs - 8) - 1) / (bs - 8), nsb = (h.ii + (bs - 4) - 1) / (bs - 4);
  const int nca = chunks(cus * (bs == 64 ? 8 : 160 * 1024 / (37 * (bs + 4) * 8)), nsa, c->momtum_chunks_a);
  const int ncb = chunks(cus * 8 * 64 / bs, nsb, c->momtum_chunks_b);
  if (bs == 64) launch_marches<64>(c, m, n, mm, nn, nca, ncb);
  else if (bs == 128) launch_marches<128>(c, m, n, mm, nn, nca, ncb);
  else if (bs == 256) launch_marches<256>(c, m, n, mm, nn, nca, ncb);
  else return ctx_fail(c, "momtum: momtum_bs must be 64, 128 or 256");
  hipLaunchKernelGGL(k_mom_column_from, plane_grid(h, 2, 64), dim3(64), 0, c->stream, c->d, m, mm, nn);
  HIPCHK(c, hipGetLastError());
  return 0;
}
