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
#include <type_traits>

// kk-level work-space slots of the fused path
enum { MF_VISU, MF_VISV, MF_UM, MF_UN, MF_VM, MF_VN, MF_NSLOT };
#define S2_DRAG 3      // 2-D work plane (written by k_mom_drag, stage_momtum.hip)
#define S2_QUM 4       // four 2-D work planes (k_mom_qplanes, stage_momtum.hip): ubflxs_p*tsfac/(pbu*scuy) at level m, the same for v, both at level n
#define WK2V(w) ((gcd_t) * (double *const volatile CONST_AS *)&Vp->wk2d + (size_t)(w) * np)      // (like GFV below: a scalar load; as a reference cast it was a vector load + v_readfirstlane in front of the step's loads)

// Field pointers come out of the DevView in memory, so the compiler cannot know their address space and would emit
// flat loads -- which count on lgkmcnt as well as vmcnt, so that every LDS wait (and the wait in front of every
// barrier) also drains the global loads in flight.  Cast to the global address space: global_load, vmcnt only.
// (gcd_t, gd_t, gci_t: blomgpu_internal.h -- since round 6 every kernel of the library gets its pointers that way)
#define GLOBAL_AS BLOM_GAS
#define GF(V, id) ((gcd_t)(V).f[id])

// chunk-major work item of this workgroup; XCD x (blockIdx % 8) walks a contiguous eighth of the items so that the
// workgroups of one XCD share the 2-D coefficient rows of their chunk in its L2
// `list` (round 6): the launch covers only the (chunk, strip) pairs it names, chunk * nstrip + strip in ascending order -- the viscous
// march's all-wet strips and the others are two launches of two kernels (launch_marches)
__device__ inline void march_item(int kk, int jj, int nchunk_, int nstrip, int &k, int &ja, int &jb, int &strip, const int *list = nullptr) {
  const bool chunk_major = nchunk_ < 0;
  const int nchunk = chunk_major ? -nchunk_ : nchunk_;
  const unsigned nitem = gridDim.x, lin = blockIdx.x;
  const unsigned xq = lin & 7u, sq = lin >> 3, q = nitem >> 3, rr = nitem & 7u;
  unsigned item = xq * q + (xq < rr ? xq : rr) + sq;
  if (list) {                     // (chunk-major order of the listed pairs: the pair of item / kk, layer item % kk)
    const int e = list[item / kk];
    k = item % kk;
    strip = e % nstrip;
    const int ch = e / nstrip, rows = (jj + nchunk - 1) / nchunk;
    ja = 1 + ch * rows;
    jb = ja + rows - 1 < jj ? ja + rows - 1 : jj;
    return;
  }
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
// Addressing: a lane's position in a plane is a 32-bit BYTE offset (planes and whole fields are < 4 GB), added to the
// field's uniform base pointer by the load itself (global_load ... v_off, s[base:base+1] offset:imm): the i-neighbours
// cost nothing (immediate offsets), a row's offset is computed once per step and shared by every field read on that
// row.  With 64-bit per-load indices the march spent ~70-110 VALU instructions per step on address arithmetic.
typedef const char GLOBAL_AS *gcc_t;
typedef char GLOBAL_AS *gc_t;
template <int IMM = 0> __device__ inline double ldo(gcd_t b, unsigned o) { return *(gcd_t)((gcc_t)b + o + IMM); }
template <int IMM = 0> __device__ inline int ldoi(gci_t b, unsigned o) { return *(gci_t)((gcc_t)b + o + IMM); }
__device__ inline void sto(gd_t b, unsigned o, double v) { *(gd_t)((gc_t)b + o) = v; }

// LDS rings: the rings of one depth share their row slots -- ring X, slot q, lane l at [q][X][l] -- so that a step
// computes ONE lane pointer per depth and row (9 in all) and every access is that pointer plus a compile-time
// offset (ring, i-neighbour), i.e. a ds_read/ds_write with an immediate offset.  Slot of row r: (r + 16 D) % D.
#define RG(p, X, dl) (p)[(X) * NP + (dl)]
// The marches read ~50 arrays, and 50 base pointers do not fit the 102 SGPRs of a wavefront: kept live across the loop
// they were spilled into VGPR lanes (v_writelane / v_readlane + hazard nops, ~100-200 instructions per step).  The
// pointers of the 2-D coefficient planes are therefore fetched again at the top of every step -- one scalar load from
// the DevView each, through the constant address space, volatile so that it stays inside the loop.
#ifdef BLOM_HOSTEMU
#define CONST_AS
#else
#define CONST_AS __attribute__((address_space(4)))
#endif
#define GFV(id) ((gcd_t) * (double *const volatile CONST_AS *)&Vp->f.p_[id])
// AWM: 2 = both bodies of a step, the mask-free one where the strip's stencil window holds no land (the production form); 1 = the mask-free
// body alone -- a TIMING EXPERIMENT (option mom_force_aw, round 6): what a launch list of all-wet strips of its own would run at; its
// results are wrong next to land and it is never used by a test or the bench's timed path unless asked for
template <int BS, int AWM = 2>
__global__ __launch_bounds__(BS) void k_mom_visc_march(const DevView *__restrict__ Vp, int m, int n, int mm, int nn, int nchunk, int nstrip, const int *list) {
  const DevView &V = *Vp;
  HIP_DYNAMIC_SHARED(double, lds)
  const int l = threadIdx.x, ni = V.ni, ii = V.ii, jj = V.jj, kk = V.kk;
  int k, ja, jb, strip;
  march_item(kk, jj, nchunk, nstrip, k, ja, jb, strip, list);
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
  [[maybe_unused]] const size_t om = (size_t)(m - 1) * np, on = (size_t)(n - 1) * np;
  // scalars of the step in registers: left in the DevView they would be re-loaded inside the loop (the kernel stores
  // to memory the compiler cannot tell apart from it)
  [[maybe_unused]] const double tsfac = V.P.dlt / V.P.delt1;
  const double mdv2hi = V.P.mdv2hi, mdv2lo = V.P.mdv2lo, mdv4hi = V.P.mdv4hi, mdv4lo = V.P.mdv4lo;
  const double vsc2hi = V.P.vsc2hi, vsc2lo = V.P.vsc2lo, vsc4hi = V.P.vsc4hi, vsc4lo = V.P.vsc4lo;
  const gci_t mpk = (gci_t)V.m[I_mpack];
  // lanes outside every sweep's i-range (i < -1, i > ii+2) and rows outside the array read a neighbour's data instead,
  // which no sweep uses
  const int xl = x < 2 ? 2 : (x > ni - 3 ? ni - 3 : x);
  const unsigned x8 = (unsigned)xl * 8u, ni8 = (unsigned)ni * 8u;
  auto roff = [&](int r) {           // byte offset of (xl, row r clamped into the array) in a plane of doubles
    const int rc = r < -2 ? -2 : (r > jj + 3 ? jj + 3 : r);
    return x8 + ni8 * (unsigned)(rc + NBDY - 1);
  };

  constexpr int NP = BS + 4;
  enum { R_UTN, R_VTN, R_DL2U, R_VS2V, R_VS4V, N4 };       // depth 4
  enum { R_VIB, R_DL2V, R_VS2U, R_VS4U, N3 };              // depth 3
  enum { R_D1, R_D2, N2 };                                 // depth 2
  double *const l4 = lds + 2 + l;                          // two pad columns on either side: l-1 .. l+2 of the edge lanes stay inside a row
  double *const l3 = l4 + 4 * N4 * NP, *const l2 = l3 + 3 * N3 * NP, *const ufl1 = l2 + 2 * N2 * NP;
  // fields that are only ever read at the lane that wrote them live in registers: value of row s-1 (written this
  // step by W), s-2, s-3 (read by U); ujb of row s-2 is read by V; vflux1 of rows s-3, s-4
  double wja1 = 0., wja2 = 0., wja3 = 0., wjb1 = 0., wjb2 = 0., wjb3 = 0., uja1 = 0., uja2 = 0., uja3 = 0.;
  double ujb1 = 0., ujb2 = 0., ujb3 = 0., wia1 = 0., wia2 = 0., wia3 = 0., wib1 = 0., wib2 = 0., wib3 = 0.;
  double via1 = 0., via2 = 0., via3 = 0., vfl3 = 0., vfl4 = 0.;

  const gcd_t f_u = GF(V, F_u) + okn, f_v = GF(V, F_v) + okn;
  const gcd_t f_pbum = GF(V, F_pbu) + om, f_pbvm = GF(V, F_pbv) + om;
  const gcd_t f_pu0 = GF(V, F_pu) + ok, f_pv0 = GF(V, F_pv) + ok;      // the interface below: one plane further on
  const unsigned np8 = (unsigned)np * 8u;
  const gcd_t f_dpu = GF(V, F_dpu) + okm, f_dpv = GF(V, F_dpv) + okm;
  const gd_t o_visu = (gd_t)WK(V, MF_VISU) + ok, o_visv = (gd_t)WK(V, MF_VISV) + ok;

  // inputs of the first sweep, one step ahead
  struct TIn { int mk; double u, qu, v, qv; };
  auto load_t = [&](int r) {
    const unsigned o = roff(r);
    const gcd_t qn = WK2V(S2_QUM + 2);
    TIn t;
    t.mk = ldoi(mpk, o >> 1);
    t.u = ldo(f_u, o); t.qu = ldo(qn, o);
    t.v = ldo(f_v, o); t.qv = ldo(qn, o + np8);
    return t;
  };
  TIn tc = load_t(ja - 2);
  // the packed masks of the rows behind the first sweep are carried along instead of being read again at every lag
  // (a row's word serves W, V at lag 1, S at lag 2, F, U at lag 3 and their j-neighbours)
  int mk1 = ldoi(mpk, roff(ja - 3) >> 1), mk2 = ldoi(mpk, roff(ja - 4) >> 1), mk3 = ldoi(mpk, roff(ja - 5) >> 1), mk4 = ldoi(mpk, roff(ja - 6) >> 1);
  int q3 = (ja - 3 + 48) % 3;        // depth-3 slot of row s-1

  // All-wet fast path (round 5).  Almost every strip-row of an ocean grid has no land in its stencil window (the channel: 106 080 of
  // 106 496 points are wet), and then every mask test of the step is true and every select takes its first operand: the step's body
  // exists twice, once with the masks as data and once with them compiled out (AW), and a wave-uniform branch picks per step.  The
  // flag: all 64 lanes' packed mask words of the rows s-4 .. s are 15 -- the words of a row are tested when the row is loaded (one
  // v_cmp + s_cmp per step) and kept as one bit of a shift register.  The lanes next to the strip (x - 1 of lane 0, x + 1, x + 2 of
  // lane 63) are not tested: they only reach the strip's outer 4 + 4 lanes, which own no output (the reach argument of the strip
  // layout).  Same bits by construction: a select on a true mask returns the operand the masked form returns.
#ifdef BLOM_HOSTEMU
#define WAVE_ALL(p) false            // (the host emulation runs lanes one after the other: it takes the masked form)
#else
#define WAVE_ALL(p) (__all(p) != 0)
#endif
#define MUa(m) (AW || MU(m))
#define MVa(m) (AW || MV(m))
#define MPa(m) (AW || MP(m))
#define MQa(m) (AW || MQ(m))
  unsigned awbits = (WAVE_ALL((tc.mk & 15) == 15) ? 1u : 0u) | (WAVE_ALL((mk1 & 15) == 15) ? 2u : 0u) | (WAVE_ALL((mk2 & 15) == 15) ? 4u : 0u) |
                    (WAVE_ALL((mk3 & 15) == 15) ? 8u : 0u) | (WAVE_ALL((mk4 & 15) == 15) ? 16u : 0u);
  const bool aw_on = V.P.allwet != 0;
  for (int s = ja - 2; s <= jb + 3; s++) {
   auto step = [&](auto awc) {
    constexpr bool AW = decltype(awc)::value;
    // ================= loads of this step =================
    const TIn tn = load_t(s + 1);
    const gcd_t scuy = GFV(F_scuy), scvx = GFV(F_scvx), scvy = GFV(F_scvy), scux = GFV(F_scux);
    const gcd_t scq2i = GFV(F_scq2i), scp2i = GFV(F_scp2i), difwgt = GFV(F_difwgt);
    const gcd_t f_difmxp = GFV(F_difmxp), f_difmxq = GFV(F_difmxq), f_scpy = GFV(F_scpy), f_scpx = GFV(F_scpx);
    const gcd_t f_scqx = GFV(F_scqx), f_scqy = GFV(F_scqy), f_scu2 = GFV(F_scu2), f_scv2 = GFV(F_scv2);
    gd_t o_utotn = nullptr, o_vtotn = nullptr;
    if (k == kk - 1) { o_utotn = (gd_t)GFV(F_utotn); o_vtotn = (gd_t)GFV(F_vtotn); }
    // W, V (row s-1)
    const unsigned ow = roff(s - 1), own_ = ow + ni8, ows = ow - ni8;
    const int w_m = mk1, w_mw = ldoi<-4>(mpk, ow >> 1), w_ms = mk2;
    const double w_pu1 = ldo(f_pu0, ow + np8), w_pu0 = ldo(f_pu0, ow), w_pbua = ldo(f_pbum, ows), w_pbub = ldo(f_pbum, own_);
    const double w_pv1 = ldo(f_pv0, ow + np8), w_pv0 = ldo(f_pv0, ow), w_pbva = ldo<-8>(f_pbvm, ow), w_pbvb = ldo<8>(f_pbvm, ow);
    const double v_scvy = ldo(scvy, ow), v_scvyw = ldo<-8>(scvy, ow), v_scux = ldo(scux, ow), v_scuxs = ldo(scux, ows), v_scq2i = ldo(scq2i, ow);
    const double v_scuy = ldo(scuy, ow), v_scuye = ldo<8>(scuy, ow), v_scvx = ldo(scvx, ow), v_scvxn = ldo(scvx, own_), v_scp2i = ldo(scp2i, ow);
    const double sv_dw = ldo(difwgt, ow), sv_dws = ldo(difwgt, ows);
    // S at u-points (row s-2)
    const unsigned os = roff(s - 2);
    const int s_m = mk2;
    const double su_dw = ldo(difwgt, os), su_dww = ldo<-8>(difwgt, os);
    // F, U (row s-3)
    const unsigned of = roff(s - 3), ofn = of + ni8, ofs = of - ni8;
    const int f_m = mk3, f_me = ldoi<4>(mpk, of >> 1), f_me2 = ldoi<8>(mpk, of >> 1), f_mn = mk2, f_mn2 = mk1;
    const int f_mw = ldoi<-4>(mpk, of >> 1), f_ms = mk4;
    const double v_difmxp = ldo(f_difmxp, of), v_scpy = ldo(f_scpy, of), v_scpx = ldo(f_scpx, of);
    const double dpu_c = ldo(f_dpu, of), dpu_e = ldo<8>(f_dpu, of), dpu_s = ldo(f_dpu, ofs), dpu_n = ldo(f_dpu, ofn);
    const double dpv_c = ldo(f_dpv, of), dpv_n = ldo(f_dpv, ofn), dpv_w = ldo<-8>(f_dpv, of), dpv_e = ldo<8>(f_dpv, of);
    const double dmq_c = ldo(f_difmxq, of), dmq_n = ldo(f_difmxq, ofn), dmq_e = ldo<8>(f_difmxq, of);
    const double scqx_c = ldo(f_scqx, of), scqx_n = ldo(f_scqx, ofn), scqy_c = ldo(f_scqy, of), scqy_e = ldo<8>(f_scqy, of);
    const double scu2_c = ldo(f_scu2, of), scv2_c = ldo(f_scv2, of);
    // lane pointers into the ring slots of this step's rows
    double *const a0 = l4 + (s & 3) * (N4 * NP), *const a1 = l4 + ((s - 1) & 3) * (N4 * NP);        // rows s (and s-4), s-1
    double *const a2 = l4 + ((s - 2) & 3) * (N4 * NP), *const a3 = l4 + ((s - 3) & 3) * (N4 * NP);  // rows s-2, s-3
    const int q2 = q3 == 0 ? 2 : q3 - 1, q1 = q2 == 0 ? 2 : q2 - 1;
    double *const b1 = l3 + q3 * (N3 * NP), *const b2 = l3 + q2 * (N3 * NP), *const b3 = l3 + q1 * (N3 * NP);   // rows s-1 (and s-4), s-2, s-3
    double *const c1 = l2 + ((s - 1) & 1) * (N2 * NP), *const c2 = l2 + ((s - 2) & 1) * (N2 * NP);   // rows s-1, s-2

    // ---- T: total velocities at the old time level, row s (:408-431; rows -1..jj+2, i = -1..ii+2) ------------
    if (act && s >= -1 && s <= jj + 2 && i >= -1 && i <= ii + 2) {
      double un = 0., vn = 0.;
      if (MUa(tc.mk)) {
        un = tc.u + tc.qu;
        // the reference's module array utotn is left holding the last layer's values outside the interior
        if (k == kk - 1) sto(o_utotn, x8 + ni8 * (unsigned)(s + NBDY - 1), un);
      }
      if (MVa(tc.mk)) {
        vn = tc.v + tc.qv;
        if (k == kk - 1) sto(o_vtotn, x8 + ni8 * (unsigned)(s + NBDY - 1), vn);
      }
      RG(a0, R_UTN, 0) = un;
      RG(a0, R_VTN, 0) = vn;
    }
    __syncthreads();
    // ---- W: side-wall weights, auxiliary velocities, del2 fields, row s-1 (:438-472) ---------------------------
    {
      const int r = s - 1;
      if (act && r >= -1 && r <= jj + 2 && i >= 0 && i <= ii + 2) {
        double uja = 0., ujb = 0., d2u = 0.;
        if (MUa(w_m)) {
          const double den = fmax2(w_pu1 - w_pu0, EPSILP);
          const double wa = fmax2(0., fmin2(1., (w_pu1 - w_pbua) / den));
          const double wb = fmax2(0., fmin2(1., (w_pu1 - w_pbub) / den));
          const double un = RG(a1, R_UTN, 0);
          uja = (1. - wa) * RG(a2, R_UTN, 0) + wa * SLIP * un;
          ujb = (1. - wb) * RG(a0, R_UTN, 0) + wb * SLIP * un;
          d2u = un - .25 * (RG(a1, R_UTN, 1) + RG(a1, R_UTN, -1) + uja + ujb);
          wja1 = wa;
          wjb1 = wb;
        }
        uja1 = uja;
        ujb1 = ujb;
        RG(a1, R_DL2U, 0) = d2u;
      }
      if (act && r >= 0 && r <= jj + 2 && i >= -1 && i <= ii + 2) {
        double via = 0., vib = 0., d2v = 0.;
        if (MVa(w_m)) {
          const double den = fmax2(w_pv1 - w_pv0, EPSILP);
          const double wa = fmax2(0., fmin2(1., (w_pv1 - w_pbva) / den));
          const double wb = fmax2(0., fmin2(1., (w_pv1 - w_pbvb) / den));
          const double vn = RG(a1, R_VTN, 0);
          via = (1. - wa) * RG(a1, R_VTN, -1) + wa * SLIP * vn;
          vib = (1. - wb) * RG(a1, R_VTN, 1) + wb * SLIP * vn;
          d2v = vn - .25 * (RG(a0, R_VTN, 0) + RG(a2, R_VTN, 0) + via + vib);
          wia1 = wa;
          wib1 = wb;
        }
        via1 = via;
        RG(b1, R_VIB, 0) = vib;
        RG(b1, R_DL2V, 0) = d2v;
      }
    }
    __syncthreads();
    // ---- V: deformation, row s-1 (:500-507, :534-541, :549-559, :577-585) ---------------------------------------
    {
      const int r = s - 1;
      if (act && r >= 0 && r <= jj + 2 && i >= 0 && i <= ii + 2) {          // defor2 at q-points
        bool have = false;
        double d2 = 0.;
        if (MVa(w_m) && !MVa(w_mw)) { const double t = RG(a1, R_VTN, 0) * (1. - SLIP) * v_scvy; d2 = t * t * v_scq2i; have = true; }
        else if (MVa(w_mw) && !MVa(w_m)) { const double t = RG(a1, R_VTN, -1) * (1. - SLIP) * v_scvyw; d2 = t * t * v_scq2i; have = true; }
        if (MUa(w_m) && !MUa(w_ms)) { const double t = RG(a1, R_UTN, 0) * (1. - SLIP) * v_scux; d2 = t * t * v_scq2i; have = true; }
        else if (MUa(w_ms) && !MUa(w_m)) { const double t = RG(a2, R_UTN, 0) * (1. - SLIP) * v_scuxs; d2 = t * t * v_scq2i; have = true; }
        if (MQa(w_m)) {
          const double t = RG(b1, R_VIB, -1) * v_scvy - via1 * v_scvyw + ujb2 * v_scux - uja1 * v_scuxs;
          d2 = t * t * v_scq2i;
          have = true;
        }
        if (have) RG(c1, R_D2, 0) = d2;
      }
      if (act && r >= -1 && r <= jj + 1 && i >= -1 && i <= ii + 1 && MPa(w_m)) {   // defor1 at p-points
        const double t = (RG(a1, R_UTN, 1) * v_scuye - RG(a1, R_UTN, 0) * v_scuy) - (RG(a0, R_VTN, 0) * v_scvxn - RG(a1, R_VTN, 0) * v_scvx);
        RG(c1, R_D1, 0) = t * t * v_scp2i;
      }
    }
    __syncthreads();
    // ---- S: deformation dependent viscosities (:829-841 at u-points, row s-2; :988-1000 at v-points, row s-1) ----
    if (act && i >= 0 && i <= ii + 1) {
      {
        const int r = s - 2;
        if (r >= 0 && r <= jj + 1 && MUa(s_m)) {
          const double q = .5 * (su_dww + su_dw);
          const double deform = sqrt(.5 * (RG(c2, R_D1, 0) + RG(c2, R_D1, -1) + RG(c2, R_D2, 0) + RG(c1, R_D2, 0)));
          RG(b2, R_VS2U, 0) = fmax2(q * mdv2hi + (1. - q) * mdv2lo, (q * vsc2hi + (1. - q) * vsc2lo) * deform);
          RG(b2, R_VS4U, 0) = fmax2(q * mdv4hi + (1. - q) * mdv4lo, (q * vsc4hi + (1. - q) * vsc4lo) * deform);
        }
      }
      {
        const int r = s - 1;
        if (r >= 0 && r <= jj + 1 && MVa(w_m)) {
          const double q = .5 * (sv_dws + sv_dw);
          const double deform = sqrt(.5 * (RG(c1, R_D1, 0) + RG(c2, R_D1, 0) + RG(c1, R_D2, 0) + RG(c1, R_D2, 1)));
          RG(a1, R_VS2V, 0) = fmax2(q * mdv2hi + (1. - q) * mdv2lo, (q * vsc2hi + (1. - q) * vsc2lo) * deform);
          RG(a1, R_VS4V, 0) = fmax2(q * mdv4hi + (1. - q) * mdv4lo, (q * vsc4hi + (1. - q) * vsc4lo) * deform);
        }
      }
    }
    __syncthreads();
    // ---- F: longitudinal turbulent momentum fluxes at p-points, row s-3 (:860-873, :1019-1034) -----------------
    {
      const int r = s - 3;
      if (act && r >= 0 && r <= jj && i >= 0 && i <= ii && MPa(f_m)) {
        if (r >= 1 && MUa(f_m) + MUa(f_me) > 0) {
          const double dpxy = fmax2(dpu_c, ONEMM), dpib = fmax2(dpu_e, ONEMM);
          // viscosity extended one point beyond wet u-segments (:845-856), cf. ext_i
          const int m0 = MUa(f_m), m1 = MUa(f_me), m2 = MUa(f_me2);
          const double v2 = (m0 ? RG(b3, R_VS2U, 0) : (m1 ? RG(b3, R_VS2U, 1) : RG(b3, R_VS2U, -1))) + (m1 ? RG(b3, R_VS2U, 1) : (m2 ? RG(b3, R_VS2U, 2) : RG(b3, R_VS2U, 0)));
          const double v4 = (m0 ? RG(b3, R_VS4U, 0) : (m1 ? RG(b3, R_VS4U, 1) : RG(b3, R_VS4U, -1))) + (m1 ? RG(b3, R_VS4U, 1) : (m2 ? RG(b3, R_VS4U, 2) : RG(b3, R_VS4U, 0)));
          ufl1[0] = fmin2(v_difmxp, v2 * v_scpy) * hfharm(dpxy, dpib) * (RG(a3, R_UTN, 0) - RG(a3, R_UTN, 1)) +
                    fmin2(.125 * v_difmxp, v4 * v_scpy) * hfharm(dpxy, dpib) * (RG(a3, R_DL2U, 0) - RG(a3, R_DL2U, 1));
        }
        if (i >= 1 && MVa(f_m) + MVa(f_mn) > 0) {
          const double dpxy = fmax2(dpv_c, ONEMM), dpjb = fmax2(dpv_n, ONEMM);
          const int m0 = MVa(f_m), m1 = MVa(f_mn), m2 = MVa(f_mn2);
          // rows r-1 (slot of row s), r, r+1, r+2
          const double v2 = (m0 ? RG(a3, R_VS2V, 0) : (m1 ? RG(a2, R_VS2V, 0) : RG(a0, R_VS2V, 0))) + (m1 ? RG(a2, R_VS2V, 0) : (m2 ? RG(a1, R_VS2V, 0) : RG(a3, R_VS2V, 0)));
          const double v4 = (m0 ? RG(a3, R_VS4V, 0) : (m1 ? RG(a2, R_VS4V, 0) : RG(a0, R_VS4V, 0))) + (m1 ? RG(a2, R_VS4V, 0) : (m2 ? RG(a1, R_VS4V, 0) : RG(a3, R_VS4V, 0)));
          vfl3 = fmin2(v_difmxp, v2 * v_scpx) * hfharm(dpxy, dpjb) * (RG(a3, R_VTN, 0) - RG(a2, R_VTN, 0)) +
                 fmin2(.125 * v_difmxp, v4 * v_scpx) * hfharm(dpxy, dpjb) * (RG(b3, R_DL2V, 0) - RG(b2, R_DL2V, 0));
        }
      }
    }
    __syncthreads();
    // ---- U: lateral turbulent momentum fluxes and the flux divergence term, row s-3 (:879-913, :1040-1076) -----
    {
      const int r = s - 3;
      if (act && own && r >= ja && r <= jb && i >= 1 && i <= ii) {
        if (MUa(f_m)) {
          const double wja = wja3, wjb = wjb3;
          const double dpxy = fmax2(dpu_c, ONEMM);
          double dpja = fmax2(dpu_s, ONEMM);
          dpja = dpja + wja * (dpxy - dpja);
          double dpjb = fmax2(dpu_n, ONEMM);
          dpjb = dpjb + wjb * (dpxy - dpjb);
          const double v2c = RG(b3, R_VS2U, 0), v4c = RG(b3, R_VS4U, 0);
          // rows r-1 (the slot of row s-1), r+1
          const double vsc2a = MUa(f_ms) == 0 ? v2c : RG(b1, R_VS2U, 0), vsc4a = MUa(f_ms) == 0 ? v4c : RG(b1, R_VS4U, 0);
          const double vsc2b = MUa(f_mn) == 0 ? v2c : RG(b2, R_VS2U, 0), vsc4b = MUa(f_mn) == 0 ? v4c : RG(b2, R_VS4U, 0);
          const double un = RG(a3, R_UTN, 0), d2 = RG(a3, R_DL2U, 0);
          const double dl2uja = (1. - wja) * RG(a0, R_DL2U, 0) + wja * SLIP * d2;          // :594-597
          const double dl2ujb = (1. - wjb) * RG(a2, R_DL2U, 0) + wjb * SLIP * d2;
          const double uflux2 = fmin2(dmq_c, (v2c + vsc2a) * scqx_c) * hfharm(dpja, dpxy) * (uja3 - un) +
                                fmin2(.125 * dmq_c, (v4c + vsc4a) * scqx_c) * hfharm(dpja, dpxy) * (dl2uja - d2);
          const double uflux3 = fmin2(dmq_n, (v2c + vsc2b) * scqx_n) * hfharm(dpjb, dpxy) * (un - ujb3) +
                                fmin2(.125 * dmq_n, (v4c + vsc4b) * scqx_n) * hfharm(dpjb, dpxy) * (d2 - dl2ujb);
          sto(o_visu, of, (ufl1[0] - ufl1[-1] + uflux3 - uflux2) / (scu2_c * fmax2(dpu_c, ONEMM)));
        }
        if (MVa(f_m)) {
          const double wia = wia3, wib = wib3;
          const double dpxy = fmax2(dpv_c, ONEMM);
          double dpia = fmax2(dpv_w, ONEMM);
          dpia = dpia + wia * (dpxy - dpia);
          double dpib = fmax2(dpv_e, ONEMM);
          dpib = dpib + wib * (dpxy - dpib);
          const double vs2 = RG(a3, R_VS2V, 0), vs4 = RG(a3, R_VS4V, 0);
          const double vsc2a = MVa(f_mw) == 0 ? vs2 : RG(a3, R_VS2V, -1), vsc4a = MVa(f_mw) == 0 ? vs4 : RG(a3, R_VS4V, -1);
          const double vsc2b = MVa(f_me) == 0 ? vs2 : RG(a3, R_VS2V, 1), vsc4b = MVa(f_me) == 0 ? vs4 : RG(a3, R_VS4V, 1);
          const double vn = RG(a3, R_VTN, 0), d2 = RG(b3, R_DL2V, 0);
          const double dl2via = (1. - wia) * RG(b3, R_DL2V, -1) + wia * SLIP * d2;          // :602-605
          const double dl2vib = (1. - wib) * RG(b3, R_DL2V, 1) + wib * SLIP * d2;
          const double vflux2 = fmin2(dmq_c, (vs2 + vsc2a) * scqy_c) * hfharm(dpia, dpxy) * (via3 - vn) +
                                fmin2(.125 * dmq_c, (vs4 + vsc4a) * scqy_c) * hfharm(dpia, dpxy) * (dl2via - d2);
          const double vflux3 = fmin2(dmq_e, (vs2 + vsc2b) * scqy_e) * hfharm(dpib, dpxy) * (vn - RG(b3, R_VIB, 0)) +
                                fmin2(.125 * dmq_e, (vs4 + vsc4b) * scqy_e) * hfharm(dpib, dpxy) * (d2 - dl2vib);
          sto(o_visv, of, (vfl3 - vfl4 + vflux3 - vflux2) / (scv2_c * fmax2(dpv_c, ONEMM)));
        }
      }
    }
    __syncthreads();
    wja3 = wja2; wja2 = wja1; wjb3 = wjb2; wjb2 = wjb1; uja3 = uja2; uja2 = uja1; ujb3 = ujb2; ujb2 = ujb1;
    wia3 = wia2; wia2 = wia1; wib3 = wib2; wib2 = wib1; via3 = via2; via2 = via1; vfl4 = vfl3;
    mk4 = mk3; mk3 = mk2; mk2 = mk1; mk1 = tc.mk;
    tc = tn;
    q3 = q3 == 2 ? 0 : q3 + 1;
    awbits = (awbits << 1) | (WAVE_ALL((tn.mk & 15) == 15) ? 1u : 0u);
   };
   if (AWM == 1) step(std::true_type{});
   else if (aw_on && (awbits & 31u) == 31u) step(std::true_type{});
   else step(std::false_type{});
  }
#undef MUa
#undef MVa
#undef MPa
#undef MQa
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
  [[maybe_unused]] const size_t om = (size_t)(m - 1) * np, on = (size_t)(n - 1) * np;
  const double delt1 = V.P.delt1, tsfac = V.P.dlt / V.P.delt1, cutoff = ONEM, thkbop = THKBOT * ONEM;
  const double wuv1 = V.P.wuv1, wuv2 = V.P.wuv2;
  const int mommth = V.P.mommth;
  const bool last_chunk = jb == jj;
  const gci_t mpk = (gci_t)V.m[I_mpack];
  const int xl = x < 2 ? 2 : (x > ni - 3 ? ni - 3 : x);
  const unsigned x8 = (unsigned)xl * 8u, ni8 = (unsigned)ni * 8u;
  auto roff = [&](int r) {           // byte offset of (xl, row r clamped into the array) in a plane of doubles
    const int rc = r < -2 ? -2 : (r > jj + 3 ? jj + 3 : r);
    return x8 + ni8 * (unsigned)(rc + NBDY - 1);
  };

  // rings (layout as in k_mom_visc_march): depth 4: UFX (+ UHMN, UHMX), depth 3: UTM, VTM, VFX, DPMX, KE (+ VHMN, VHMX), depth 2: PV
  constexpr int NP = BS + 4;
  enum { R_UFX, R_UHMN, R_UHMX };
  enum { R_UTM, R_VTM, R_VFX, R_DPMX, R_KE, R_VHMN, R_VHMX };
  constexpr int N4 = ENEDIS ? 3 : 1, N3 = ENEDIS ? 7 : 5;
  double *const l4 = lds + 2 + l;
  double *const l3 = l4 + 4 * N4 * NP, *const l2 = l3 + 3 * N3 * NP;

  // 3-D fields read at both time levels (u, v, pgfx, pgfy) and the 2-D ones with two time levels (ubflxs_p, vbflxs_p, pbu,
  // pbv): ONE base pointer each, at the lower of the two levels, the level going into the lane's byte offset
  const size_t olo = okm < okn ? okm : okn;
  const unsigned dkm = (unsigned)((okm - olo) * 8), dkn = (unsigned)((okn - olo) * 8);
  const unsigned np8 = (unsigned)np * 8u;
  const gcd_t dp = GF(V, F_dp) + okm, f_u = GF(V, F_u) + olo, f_v = GF(V, F_v) + olo;
  const gcd_t f_pbum = GF(V, F_pbu) + om, f_pbvm = GF(V, F_pbv) + om;
  const gcd_t f_dpu = GF(V, F_dpu) + okm, f_dpv = GF(V, F_dpv) + okm;
  const gcd_t p0 = GF(V, F_p) + ok;                 // p1: one plane further on
  const gcd_t pgfx = GF(V, F_pgfx) + olo, pgfx_o = GF(V, F_pgfx_o) + ok;
  const gcd_t pgfy = GF(V, F_pgfy) + olo, pgfy_o = GF(V, F_pgfy_o) + ok;
  const gcd_t dpuold = GF(V, F_dpuold) + ok, dpvold = GF(V, F_dpvold) + ok;
  const bool hybrid = V.P.vcoord_tag != 1;
  const gcd_t visu = (gcd_t)WK(V, MF_VISU) + ok, visv = (gcd_t)WK(V, MF_VISV) + ok;
  const gd_t o_um = (gd_t)WK(V, MF_UM) + ok, o_un = (gd_t)WK(V, MF_UN) + ok, o_vm = (gd_t)WK(V, MF_VM) + ok, o_vn = (gd_t)WK(V, MF_VN) + ok;
  const gd_t o_absvor = (gd_t)V.f[F_absvor] + ok, o_dpvor = (gd_t)V.f[F_dpvor] + ok;

  // Inputs of the first sweep are loaded one step ahead; what the later sweeps need of the same rows (the wet masks
  // and dp of rows s-1, s-2 for the vorticity, u, v, dpu, dpv and the 2-D coefficients of row s-2 for the update)
  // is carried in registers from step to step instead of being read again: the second read of a row, two march
  // steps later and with ~2000 waves streaming through the same L2, mostly missed it.
  struct TIn { int mk, mkw; double dc, dw, u, qu, dpu, v, qv, dpv; };
  auto load_t = [&](int r) {
    const unsigned o = roff(r), o3 = o + dkm;
    const gcd_t qm = WK2V(S2_QUM);
    TIn t;
    t.mk = ldoi(mpk, o >> 1); t.mkw = ldoi<-4>(mpk, o >> 1);
    t.dc = ldo(dp, o); t.dw = ldo<-8>(dp, o);
    t.u = ldo(f_u, o3); t.qu = ldo(qm, o); t.dpu = ldo(f_dpu, o);
    t.v = ldo(f_v, o3); t.qv = ldo(qm, o + np8); t.dpv = ldo(f_dpv, o);
    return t;
  };
  struct VIn { double scvy, scvyw, scux, scuxs, scq2i, cor, scu2, scu2e, scv2, scv2n, scp2; };
  auto load_v = [&](int r) {
    const unsigned o = roff(r);
    const gcd_t scvy = GFV(F_scvy), scux = GFV(F_scux), scq2i = GFV(F_scq2i), corioq = GFV(F_corioq);
    const gcd_t scu2 = GFV(F_scu2), scv2 = GFV(F_scv2), scp2 = GFV(F_scp2);
    VIn t;
    t.scvy = ldo(scvy, o); t.scvyw = ldo<-8>(scvy, o); t.scux = ldo(scux, o); t.scuxs = ldo(scux, o - ni8); t.scq2i = ldo(scq2i, o);
    t.cor = ldo(corioq, o);
    t.scu2 = ldo(scu2, o); t.scu2e = ldo<8>(scu2, o); t.scv2 = ldo(scv2, o); t.scv2n = ldo(scv2, o + ni8); t.scp2 = ldo(scp2, o);
    return t;
  };
  TIn tpp = load_t(ja - 3), tp = load_t(ja - 2), tc = load_t(ja - 1);      // rows s-2, s-1, s
  VIn vc = load_v(ja - 2);                                                 // row s-1
  double pr_p0 = ldo(p0, roff(ja - 4)), pr_p1 = ldo(p0, roff(ja - 4) + np8), pr_drag = ldo((gcd_t)WK2(V, S2_DRAG), roff(ja - 4));   // row s-3
  int q0 = (ja - 1 + 48) % 3;        // depth-3 slot of row s

  for (int s = ja - 1; s <= jb + 2; s++) {
    // ================= loads of this step =================
    const TIn tn = load_t(s + 1);
    const gcd_t scux = GFV(F_scux), scvy = GFV(F_scvy);
    const gcd_t ubcors = GFV(F_ubcors_p), vbcors = GFV(F_vbcors_p), scuxi = GFV(F_scuxi), scvyi = GFV(F_scvyi);
    const gcd_t drag = WK2V(S2_DRAG);
    gcd_t taux = nullptr, tauy = nullptr, munl = nullptr, mvnl = nullptr;      // the wind stress acts on the top layer (isopyc_bulkml) / on every layer
    if (hybrid || k == 0) { taux = GFV(F_taux); tauy = GFV(F_tauy); }
    if (hybrid) { munl = GFV(F_mu_nonloc) + ok; mvnl = GFV(F_mv_nonloc) + ok; }
    // V (row s-1): its 2-D coefficients were loaded a step ahead as well (the sweep follows the short first one at once: loaded in
    // this step they were waited for, 36 % of the wavefronts' time); this step loads row s for the next
    const unsigned ov = roff(s - 1);
    const VIn vnx = load_v(s);
    const int v_m = tp.mk, v_mw = tp.mkw, v_ms = tpp.mk;
    const double v_scvy = vc.scvy, v_scvyw = vc.scvyw, v_scux = vc.scux, v_scuxs = vc.scuxs, v_scq2i = vc.scq2i;
    const double v_dc = tp.dc, v_dw = tp.dw, v_ds = tpp.dc, v_dsw = tpp.dw, v_cor = vc.cor;
    const double v_scu2 = vc.scu2, v_scu2e = vc.scu2e, v_scv2 = vc.scv2, v_scv2n = vc.scv2n, v_scp2 = vc.scp2;
    // U (row s-2)
    const unsigned ou = roff(s - 2), ou3 = ou + dkn, ou3m = ou + dkm, oup1 = ou + np8;
    const gcd_t qn = WK2V(S2_QUM + 2);
    const int u_m = tpp.mk;
    const double u_drag = ldo(drag, ou), u_dragw = ldo<-8>(drag, ou), u_drags = pr_drag;
    const double u_p0 = ldo(p0, ou), u_p0w = ldo<-8>(p0, ou), u_p0s = pr_p0, u_p1 = ldo(p0, oup1), u_p1w = ldo<-8>(p0, oup1), u_p1s = pr_p1;
    const double u_dpu = tpp.dpu, u_pbum = ldo(f_pbum, ou), u_ukm = tpp.u, u_ukn = ldo(f_u, ou3), u_qun = ldo(qn, ou);
    const double u_pgm = ldo(pgfx, ou3m), u_pgo = ldo(pgfx_o, ou), u_pgn = ldo(pgfx, ou3), u_dpuold = ldo(dpuold, ou);
    const double u_ubcors = ldo(ubcors, ou), u_scuxi = ldo(scuxi, ou), u_visu = ldo(visu, ou);
    const double u_dpv = tpp.dpv, u_pbvm = ldo(f_pbvm, ou), u_vkm = tpp.v, u_vkn = ldo(f_v, ou3), u_qvn = ldo(qn, ou + np8);
    const double u_pgym = ldo(pgfy, ou3m), u_pgyo = ldo(pgfy_o, ou), u_pgyn = ldo(pgfy, ou3), u_dpvold = ldo(dpvold, ou);
    const double u_vbcors = ldo(vbcors, ou), u_scvyi = ldo(scvyi, ou), u_visv = ldo(visv, ou);
    // the first sweep's neighbours to the south come from the row loaded a step earlier
    const int t_mks = tp.mk;
    const double t_ds = tp.dc, t_dsw = tp.dw;
    // lane pointers into the ring slots of this step's rows
    double *const a0 = l4 + (s & 3) * (N4 * NP), *const a2 = l4 + ((s - 2) & 3) * (N4 * NP), *const a3 = l4 + ((s - 3) & 3) * (N4 * NP);   // rows s, s-2, s-3
    const int q1 = q0 == 0 ? 2 : q0 - 1, q2 = q1 == 0 ? 2 : q1 - 1;
    double *const b0 = l3 + q0 * (N3 * NP), *const b1 = l3 + q1 * (N3 * NP), *const b2 = l3 + q2 * (N3 * NP);   // rows s (and s-3), s-1, s-2
    double *const c1 = l2 + ((s - 1) & 1) * NP, *const c2 = l2 + ((s - 2) & 1) * NP;                           // PV of rows s-1, s-2

    // ---- T: total velocities at the mid time level, fluxes, dpmx, row s (:360-406; rows 0..jj+1 / dpmx 0..jj+2) ----
    if (act && s >= 0 && s <= jj + 2 && i >= 0 && i <= ii + 2) {
      double d = 8. * cutoff;
      if (MU(tc.mk)) d = fmax2(d, tc.dc + tc.dw);
      if (MU(t_mks)) d = fmax2(d, t_ds + t_dsw);
      if (MV(tc.mk)) d = fmax2(d, tc.dc + t_ds);
      if (MV(tc.mkw)) d = fmax2(d, tc.dw + t_dsw);
      RG(b0, R_DPMX, 0) = d;
      if (s <= jj + 1 && i <= ii + 1) {
        double ut = 0., uf = 0., vt = 0., vf = 0.;
        if (MU(tc.mk)) {
          ut = tc.u + tc.qu;
          uf = ut * fmax2(tc.dpu, cutoff);
        }
        if (MV(tc.mk)) {
          vt = tc.v + tc.qv;
          vf = vt * fmax2(tc.dpv, cutoff);
        }
        RG(b0, R_UTM, 0) = ut;
        RG(a0, R_UFX, 0) = uf;
        RG(b0, R_VTM, 0) = vt;
        RG(b0, R_VFX, 0) = vf;
        if (ENEDIS) {                                    // :662-715, rows 0..jj+1
          double a = 0., b = 0.;
          if (MU(tc.mk)) enedis_minmax(.5 * ut * (tc.dc + tc.dw), uf, a, b);
          RG(a0, R_UHMN, 0) = a; RG(a0, R_UHMX, 0) = b;
          a = 0.; b = 0.;
          if (MV(tc.mk)) enedis_minmax(.5 * vt * (tc.dc + t_ds), vf, a, b);
          RG(b0, R_VHMN, 0) = a; RG(b0, R_VHMX, 0) = b;
        }
      }
    }
    __syncthreads();
    // ---- V: vorticity / potential vorticity at q-points (:477-575) and kinetic energy (:613-629), row s-1 -------
    {
      const int r = s - 1;
      if (act && r >= 1 && r <= jj + 1 && i >= 1 && i <= ii + 1) {
        // UTM rows r, r-1: b1, b2; VTM row r: b1; DPMX rows r, r-1, r+1: b1, b2, b0
        bool have = false;
        double vort = 0., dpv = 1.;
        if (MV(v_m) && !MV(v_mw)) {                      // first point of a v-segment, :479-486
          vort = RG(b1, R_VTM, 0) * (1. - SLIP) * v_scvy * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_dc + v_ds), RG(b1, R_DPMX, 0)), RG(b1, R_DPMX, 1));
          have = true;
        } else if (MV(v_mw) && !MV(v_m)) {               // one past the last point of a v-segment, :487-494
          vort = -RG(b1, R_VTM, -1) * (1. - SLIP) * v_scvyw * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_dw + v_dsw), RG(b1, R_DPMX, -1)), RG(b1, R_DPMX, 0));
          have = true;
        }
        if (MU(v_m) && !MU(v_ms)) {                      // first point (in j) of a u-segment, :513-520
          vort = -RG(b1, R_UTM, 0) * (1. - SLIP) * v_scux * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_dc + v_dw), RG(b1, R_DPMX, 0)), RG(b0, R_DPMX, 0));
          have = true;
        } else if (MU(v_ms) && !MU(v_m)) {               // one past the last point, :521-528
          vort = RG(b2, R_UTM, 0) * (1. - SLIP) * v_scuxs * v_scq2i;
          dpv = .125 * fmax2(fmax2(4. * (v_ds + v_dsw), RG(b2, R_DPMX, 0)), RG(b1, R_DPMX, 0));
          have = true;
        }
        if (MQ(v_m)) {                                   // interior (incl. promontories), :561-575
          vort = (RG(b1, R_VTM, 0) * v_scvy - RG(b1, R_VTM, -1) * v_scvyw - RG(b1, R_UTM, 0) * v_scux + RG(b2, R_UTM, 0) * v_scuxs) * v_scq2i;
          double d = fmax2(2. * (v_dc + v_dw + v_ds + v_dsw), RG(b1, R_DPMX, 0));
          d = fmax2(d, RG(b1, R_DPMX, -1));
          d = fmax2(d, RG(b1, R_DPMX, 1));
          d = fmax2(d, RG(b2, R_DPMX, 0));
          d = fmax2(d, RG(b0, R_DPMX, 0));
          dpv = .125 * d;
          have = true;
        }
        if (have) {
          const double av = vort + v_cor;
          if ((own || (x == ii + NBDY && ox1 == ii + NBDY - 1)) && ((r >= ja && r <= jb) || (last_chunk && r == jj + 1))) {
            sto(o_absvor, ov, av);
            sto(o_dpvor, ov, dpv);
          }
          c1[0] = av / dpv;
        }
      }
      if (act && r >= 0 && r <= jj && i >= 0 && i <= ii && MP(v_m)) {
        const double ue = RG(b1, R_UTM, 1), uw = RG(b1, R_UTM, 0), vn = RG(b0, R_VTM, 0), vs = RG(b1, R_VTM, 0);
        RG(b1, R_KE, 0) = .25 * (v_scu2 * (uw * uw) + v_scu2e * (ue * ue) + v_scv2 * (vs * vs) + v_scv2n * (vn * vn)) / v_scp2;
      }
    }
    __syncthreads();
    // ---- U: Coriolis/advection, stresses, pressure gradient, update of both time levels, row s-2 ----------------
    {
      const int r = s - 2;
      if (act && own && r >= ja && r <= jb && i >= 1 && i <= ii) {
        // PV rows r, r+1: c2, c1; KE rows r, r-1: b2, b0 (the slot of row s holds row s-3)
        if (MU(u_m)) {
          // VFX rows r, r+1: b2, b1
          double cau;
          if (ENEDIS) {                                                            // enedis, :771-790
            const double utm = RG(b2, R_UTM, 0);
            double t1, t2;
            const double pn = c1[0], pc = c2[0];
            if (pn * utm == 0.) t1 = pn * ((RG(b1, R_VHMX, 0) + RG(b1, R_VHMX, -1)) + (RG(b1, R_VHMN, 0) + RG(b1, R_VHMN, -1))) * .5;
            else if (pn * utm < 0.) t1 = pn * (RG(b1, R_VHMX, 0) + RG(b1, R_VHMX, -1));
            else t1 = pn * (RG(b1, R_VHMN, 0) + RG(b1, R_VHMN, -1));
            if (pc * utm == 0.) t2 = pc * ((RG(b2, R_VHMX, 0) + RG(b2, R_VHMX, -1)) + (RG(b2, R_VHMN, 0) + RG(b2, R_VHMN, -1))) * .5;
            else if (pc * utm < 0.) t2 = pc * (RG(b2, R_VHMX, 0) + RG(b2, R_VHMX, -1));
            else t2 = pc * (RG(b2, R_VHMN, 0) + RG(b2, R_VHMN, -1));
            cau = .25 * (t1 + t2);
          } else if (mommth == 0)
            cau = .125 * (RG(b2, R_VFX, 0) + RG(b1, R_VFX, 0) + RG(b2, R_VFX, -1) + RG(b1, R_VFX, -1)) * (c2[0] + c1[0]);
          else
            cau = .25 * ((RG(b2, R_VFX, 0) + RG(b2, R_VFX, -1)) * c2[0] + (RG(b1, R_VFX, 0) + RG(b1, R_VFX, -1)) * c1[0]);
          // wind stress (isopyc_bulkml: top layer only), :919-936
          double stress = 0.;
          if (hybrid)                    // the other vertical coordinates: the stress spread by the non-local fractions, :937-946
            stress = -(ldo(munl, ou) - ldo(munl, oup1)) * ldo(taux, ou) * GRAV * ldo(scux, ou) / fmax2(ONEMM, u_dpu);
          else if (k == 0) stress = -2. * ldo(taux, ou) * GRAV * ldo(scux, ou) / (ldo(p0, oup1) + ldo<-8>(p0, oup1));
          const double pbu = u_pbum;
          const double ptopl = .5 * (fmin2(pbu, u_p0) + fmin2(pbu, u_p0w));
          const double pbotl = .5 * (fmin2(pbu, u_p1) + fmin2(pbu, u_p1w));
          const double q = .5 * (u_drag + u_dragw) * (fmax2(pbu - thkbop, pbotl) - fmax2(pbu - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                           fmax2(u_dpu, ONEMM);
          const double ukm = u_ukm, ukn = u_ukn;
          const double un = ukn + u_qun;                                            // utotn, :408-414
          const double botstr = -un * q / (1. + delt1 * q);
          const double pgf = (1. - 2. * WPGF) * u_pgm + WPGF * (u_pgo + u_pgn);
          sto(o_um, ou, ukm * (wuv1 * u_dpu + ONEMM) + ukn * wuv2 * u_dpuold);
          const double ubrhs = u_ubcors * tsfac;                                    // :302
          sto(o_un, ou, ukn + delt1 * (-u_scuxi * (-pgf + stress + (RG(b2, R_KE, 0) - RG(b2, R_KE, -1))) + cau - ubrhs + botstr - u_visu));
        }
        if (MV(u_m)) {
          // UFX rows r, r-1: a2, a3
          double cav;
          if (ENEDIS) {                                                            // enedis, :793-812
            const double vtm = RG(b2, R_VTM, 0);
            double t1, t2;
            const double pe = c2[1], pc = c2[0];
            if (pe * vtm == 0.) t1 = pe * ((RG(a2, R_UHMX, 1) + RG(a3, R_UHMX, 1)) + (RG(a2, R_UHMN, 1) + RG(a3, R_UHMN, 1))) * .5;
            else if (pe * vtm > 0.) t1 = pe * (RG(a2, R_UHMX, 1) + RG(a3, R_UHMX, 1));
            else t1 = pe * (RG(a2, R_UHMN, 1) + RG(a3, R_UHMN, 1));
            if (pc * vtm == 0.) t2 = pc * ((RG(a2, R_UHMX, 0) + RG(a3, R_UHMX, 0)) + (RG(a2, R_UHMN, 0) + RG(a3, R_UHMN, 0))) * .5;
            else if (pc * vtm > 0.) t2 = pc * (RG(a2, R_UHMX, 0) + RG(a3, R_UHMX, 0));
            else t2 = pc * (RG(a2, R_UHMN, 0) + RG(a3, R_UHMN, 0));
            cav = -.25 * (t1 + t2);
          } else if (mommth == 0)
            cav = -.125 * (RG(a2, R_UFX, 0) + RG(a2, R_UFX, 1) + RG(a3, R_UFX, 0) + RG(a3, R_UFX, 1)) * (c2[0] + c2[1]);
          else
            cav = -.25 * ((RG(a2, R_UFX, 0) + RG(a3, R_UFX, 0)) * c2[0] + (RG(a2, R_UFX, 1) + RG(a3, R_UFX, 1)) * c2[1]);
          double stress = 0.;
          if (hybrid)                    // :1100-1109
            stress = -(ldo(mvnl, ou) - ldo(mvnl, oup1)) * ldo(tauy, ou) * GRAV * ldo(scvy, ou) / fmax2(ONEMM, u_dpv);
          else if (k == 0) stress = -2. * ldo(tauy, ou) * GRAV * ldo(scvy, ou) / (ldo(p0, oup1) + ldo(p0, oup1 - ni8));
          const double pbv = u_pbvm;
          const double ptopl = .5 * (fmin2(pbv, u_p0) + fmin2(pbv, u_p0s));
          const double pbotl = .5 * (fmin2(pbv, u_p1) + fmin2(pbv, u_p1s));
          const double q = .5 * (u_drag + u_drags) * (fmax2(pbv - thkbop, pbotl) - fmax2(pbv - thkbop, fmin2(ptopl, pbotl - ONEMM))) /
                           fmax2(u_dpv, ONEMM);
          const double vkm = u_vkm, vkn = u_vkn;
          const double vn = vkn + u_qvn;                                            // vtotn, :424-430
          const double botstr = -vn * q / (1. + delt1 * q);
          const double pgf = (1. - 2. * WPGF) * u_pgym + WPGF * (u_pgyo + u_pgyn);
          sto(o_vm, ou, vkm * (wuv1 * u_dpv + ONEMM) + vkn * wuv2 * u_dpvold);
          const double vbrhs = u_vbcors * tsfac;                                    // :307
          sto(o_vn, ou, vkn + delt1 * (-u_scvyi * (-pgf + stress + (RG(b2, R_KE, 0) - RG(b0, R_KE, 0))) + cav - vbrhs + botstr - u_visv));
        }
      }
    }
    __syncthreads();
    pr_p0 = u_p0; pr_p1 = u_p1; pr_drag = u_drag;
    tpp = tp; tp = tc; tc = tn;
    vc = vnx;
    q0 = q0 == 2 ? 0 : q0 + 1;
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
  gd_t u = isv ? V.f[F_v] : V.f[F_u];
  gcd_t sm = WK(V, isv ? MF_VM : MF_UM), sn = WK(V, isv ? MF_VN : MF_UN);
  gcd_t dpu = isv ? V.f[F_dpv] : V.f[F_dpu], dpuold = isv ? V.f[F_dpvold] : V.f[F_dpuold];
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
  gd_t pun = (isv ? V.f[F_pv] : V.f[F_pu]) + c;
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

// One wavefront per (chunk, strip) pair of the viscous march: 1 where every packed mask word the march of that pair tests -- its 64 lanes
// (the march's own clamped column) on the rows ja - 6 .. jb + 4 (its clamped rows) -- is 15, i.e. where EVERY step of the pair's march
// takes the mask-free body.  Masks are constant: evaluated once per (BS, chunk count) and kept (blomgpu_ctx::mom_aw_*).
template <int BS>
__global__ __launch_bounds__(64) void k_mom_classify(const DevView *__restrict__ Vp, int nchunk, int nstrip, int *flags) {
  const DevView &V = *Vp;
  const int l = threadIdx.x, ni = V.ni, ii = V.ii, jj = V.jj;
  const int e = blockIdx.x, strip = e % nstrip, ch = e / nstrip;
  const int rows = (jj + nchunk - 1) / nchunk, ja = 1 + ch * rows, jb = ja + rows - 1 < jj ? ja + rows - 1 : jj;
  constexpr int HL = 4, HR = 4, OW = BS - HL - HR;
  const int ox0 = NBDY + strip * OW;
  bool all = ja <= jb && ox0 <= ii + NBDY - 1;
  for (int l0 = 0; l0 < BS; l0 += 64) {
    const int x = ox0 - HL + l0 + l;
    const int xl = x < 2 ? 2 : (x > ni - 3 ? ni - 3 : x);
    for (int r = ja - 6; r <= jb + 4; r++) {
      const int rc = r < -2 ? -2 : (r > jj + 3 ? jj + 3 : r);
      if ((V.m[I_mpack][xl + ni * (rc + NBDY - 1)] & 15) != 15) all = false;
    }
  }
  all = WAVE_ALL(all);
  if (l == 0) flags[e] = all ? 1 : 0;
}

// the two launch lists of the viscous march (device arrays of chunk * nstrip + strip): [0] the all-wet pairs, [1] the others
static int mom_aw_lists(blomgpu_ctx *c, int BS, int nca, int nsa, const int **la_, int *na, const int **lb_, int *nb) {
  const int key = BS * 100000 + nca * 100 + nsa;
  if (c->mom_aw_key != key) {
    // (allocations and a read-back: not while the stream is being captured into a graph -- blomgpu_step captures after four plain steps,
    // by which time the lists exist; a capture that meets them missing takes the one-kernel form)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return 1;
    if (c->mom_aw_dev) { (void)hipFree(c->mom_aw_dev); c->mom_aw_dev = nullptr; }
    const int n = nca * nsa;
    int *flags = nullptr;
    HIPCHK(c, hipMalloc((void **)&flags, sizeof(int) * n));
    ctx_sync_view(c);
    if (BS == 64) hipLaunchKernelGGL(k_mom_classify<64>, dim3(n), dim3(64), 0, c->stream, c->d, nca, nsa, flags);
    else if (BS == 128) hipLaunchKernelGGL(k_mom_classify<128>, dim3(n), dim3(64), 0, c->stream, c->d, nca, nsa, flags);
    else hipLaunchKernelGGL(k_mom_classify<256>, dim3(n), dim3(64), 0, c->stream, c->d, nca, nsa, flags);
    std::vector<int> hf(n), lists;
    HIPCHK(c, hipMemcpyAsync(hf.data(), flags, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(flags);
    for (int e = 0; e < n; e++) if (hf[e]) lists.push_back(e);
    c->mom_aw_n[0] = (int)lists.size();
    for (int e = 0; e < n; e++) if (!hf[e]) lists.push_back(e);
    c->mom_aw_n[1] = n - c->mom_aw_n[0];
    HIPCHK(c, hipMalloc((void **)&c->mom_aw_dev, sizeof(int) * (n > 0 ? n : 1)));
    HIPCHK(c, hipMemcpy(c->mom_aw_dev, lists.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    c->mom_aw_key = key;
  }
  *la_ = c->mom_aw_dev; *na = c->mom_aw_n[0];
  *lb_ = c->mom_aw_dev + c->mom_aw_n[0]; *nb = c->mom_aw_n[1];
  return 0;
}

template <int BS>
static void launch_marches(blomgpu_ctx *c, int m, int n, int mm, int nn, int nca, int ncb, int part) {
  const DevView &h = c->h;
  const int nsa = (h.ii + (BS - 8) - 1) / (BS - 8), nsb = (h.ii + (BS - 4) - 1) / (BS - 4);
  const size_t la = sizeof(double) * 37 * (BS + 4) + c->momtum_lds_pad, lb = sizeof(double) * (h.P.mommth == 2 ? 35 : 21) * (BS + 4) + c->momtum_lds_pad;
  // more than 64 KB of dynamic LDS has to be asked for
  (void)hipFuncSetAttribute((const void *)k_mom_visc_march<BS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)la);
  (void)hipFuncSetAttribute((const void *)k_mom_cor_march<BS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
  (void)hipFuncSetAttribute((const void *)k_mom_cor_march<BS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
  if (part != 2) {
    TimeScope tk(c, "k_mom_visc_march");
    (void)hipFuncSetAttribute((const void *)k_mom_visc_march<BS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)la);
    const int *l_aw = nullptr, *l_mx = nullptr;
    int n_aw = 0, n_mx = 0;
    // Two kernels, not two bodies (round 6): the (chunk, strip) pairs whose every step is all-wet run the kernel that holds the mask-free body
    // alone (149 VGPRs, 13 SGPR spills against 175 / 45 with both bodies: 243 against 285 us for the whole layer set), the pairs next to land
    // the kernel with both bodies.  The lists need chunk-major order (the listed pair is item / kk).
    const bool split = c->mom_aw_split && h.P.allwet && !c->mom_force_aw && !c->momtum_order && mom_aw_lists(c, BS, nca, nsa, &l_aw, &n_aw, &l_mx, &n_mx) == 0;
    if (c->mom_force_aw)
      hipLaunchKernelGGL((k_mom_visc_march<BS, 1>), dim3(h.kk * nca * nsa), dim3(BS), la, c->stream, ctx_view(c, VIEW_MOM_A), m, n, mm, nn, c->momtum_order ? nca : -nca, nsa, (const int *)nullptr);
    else if (split) {
      // The two launches must run SIDE BY SIDE: a march's wavefront lives ~240 us whatever the size of the launch (one after the other they
      // took 226 + 174 us against 288 for the one kernel).  The pairs next to land go to a stream of their own, forked from and joined to the
      // stream the march is on.
      const bool third = n_aw && n_mx && c->mom_aw_split == 2;
      if (third) {
        if (!c->side3) {
          if (hipStreamCreateWithFlags(&c->side3, hipStreamNonBlocking) != hipSuccess) c->side3 = nullptr;
          for (auto &e : c->ev_side3) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
        }
      }
      if (third && c->side3 && c->ev_side3[0] && c->ev_side3[1]) {
        (void)hipEventRecord(c->ev_side3[0], c->stream);
        (void)hipStreamWaitEvent(c->side3, c->ev_side3[0], 0);
        hipLaunchKernelGGL((k_mom_visc_march<BS, 2>), dim3(h.kk * n_mx), dim3(BS), la, c->side3, ctx_view(c, VIEW_MOM_A), m, n, mm, nn, -nca, nsa, l_mx);
        hipLaunchKernelGGL((k_mom_visc_march<BS, 1>), dim3(h.kk * n_aw), dim3(BS), la, c->stream, ctx_view(c, VIEW_MOM_A), m, n, mm, nn, -nca, nsa, l_aw);
        (void)hipEventRecord(c->ev_side3[1], c->side3);
        (void)hipStreamWaitEvent(c->stream, c->ev_side3[1], 0);
      } else {
        if (n_mx) hipLaunchKernelGGL((k_mom_visc_march<BS, 2>), dim3(h.kk * n_mx), dim3(BS), la, c->stream, ctx_view(c, VIEW_MOM_A), m, n, mm, nn, -nca, nsa, l_mx);
        if (n_aw) hipLaunchKernelGGL((k_mom_visc_march<BS, 1>), dim3(h.kk * n_aw), dim3(BS), la, c->stream, ctx_view(c, VIEW_MOM_A), m, n, mm, nn, -nca, nsa, l_aw);
      }
    } else
      hipLaunchKernelGGL((k_mom_visc_march<BS, 2>), dim3(h.kk * nca * nsa), dim3(BS), la, c->stream, ctx_view(c, VIEW_MOM_A), m, n, mm, nn, c->momtum_order ? nca : -nca, nsa, (const int *)nullptr);
  }
  if (part == 1) return;
  TimeScope tk(c, "k_mom_cor_march");
  if (h.P.mommth == 2)
    hipLaunchKernelGGL((k_mom_cor_march<BS, true>), dim3(h.kk * ncb * nsb), dim3(BS), lb, c->stream, ctx_view(c, VIEW_MOM_B), m, n, mm, nn, c->momtum_order ? ncb : -ncb, nsb);
  else
    hipLaunchKernelGGL((k_mom_cor_march<BS, false>), dim3(h.kk * ncb * nsb), dim3(BS), lb, c->stream, ctx_view(c, VIEW_MOM_B), m, n, mm, nn, c->momtum_order ? ncb : -ncb, nsb);
}

// the layer loop and the vertical pass of momtum; the caller (st_momtum) has done p/pu/pv, the drag and difwgt's halo.
// part 1: the viscous march alone; 2: the Coriolis march and the vertical pass; 0: all three
int st_momtum_fused_layers(blomgpu_ctx *c, int m, int n, int mm, int nn, int part) {
  const DevView &h = c->h;
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
  if (bs == 64) launch_marches<64>(c, m, n, mm, nn, nca, ncb, part);
  else if (bs == 128) launch_marches<128>(c, m, n, mm, nn, nca, ncb, part);
  else if (bs == 256) launch_marches<256>(c, m, n, mm, nn, nca, ncb, part);
  else return ctx_fail(c, "momtum: momtum_bs must be 64, 128 or 256");
  if (part != 1) hipLaunchKernelGGL(k_mom_column_from, plane_grid(h, 2, 64), dim3(64), 0, c->stream, ctx_view(c, VIEW_MOM_C), m, mm, nn);
  HIPCHK(c, hipGetLastError());
  return 0;
}
