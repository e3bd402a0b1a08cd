// hor3map_core.h -- per-column routines of the batched 1-D reconstruction / regridding /
// remapping library (the device counterpart of the reference's phy/mod_hor3map.F90).
//
// One GPU thread owns one column.  Every per-column array lives in HBM as [level][column]
// (column fastest), so that the 64 lanes of a wavefront, which walk their columns level by level
// in lock step, always touch 64 consecutive doubles.  The three reference data structures keep
// their roles and names:
//   H3Grid  <-> recon_grd_struct  (mod_hor3map.F90:153)  source grid, merge map, edge-solver coefficients
//   H3Src   <-> recon_src_struct  (:207)                 one reconstructed source field
//   H3Map   <-> remap_struct      (:242)                 segment list of one destination grid
// Arithmetic follows the reference's evaluation order operation by operation (no contraction),
// which is what makes the results bit-identical to it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

#define H3HD __host__ __device__ inline

enum {  // mod_hor3map.F90:47-57
  H3_PCM = 100, H3_PLM = 101, H3_PPM = 102, H3_PQM = 103,
  H3_NO_LIMITING = 200, H3_MONOTONIC = 201, H3_NON_OSCILLATORY = 203, H3_NON_OSCILLATORY_POSDEF = 204,
  H3_REGRID_METHOD_1 = 301, H3_REGRID_METHOD_2 = 302
};
enum {  // mod_hor3map.F90:60-83
  H3_NOERR = 0, H3_INVALID_RECON_METHOD = 1, H3_RESIZING_INITIALIZED_RCGS = 2,
  H3_NONMONOTONIC_SRC_EDGES = 3, H3_SRC_EXTENT_TOO_SMALL = 4, H3_FAILED_TO_ALLOCATE_RCGS = 5,
  H3_RECON_NOT_PREPARED = 6, H3_RESIZING_INITIALIZED_RMS = 7, H3_INCONSISTENT_GRID_RANGE = 8,
  H3_NONMONOTONIC_DST_EDGES = 9, H3_FAILED_TO_ALLOCATE_RMS = 10, H3_SRC_SIZE_MISMATCH = 11,
  H3_FAILED_TO_ALLOCATE_RCSS = 12, H3_INVALID_PLM_LIMITING = 13, H3_INVALID_PPM_LIMITING = 14,
  H3_INVALID_PQM_LIMITING = 15, H3_RECON_NOT_AVAILABLE = 16, H3_INVALID_REGRID_METHOD = 17,
  H3_GRD_SIZE_MISMATCH = 18, H3_REMAP_NOT_PREPARED = 19, H3_DST_SIZE_MISMATCH = 20,
  H3_INDEX_OUT_OF_BOUNDS = 21, H3_INCONSISTENT_RCGS = 22
};

constexpr double H3_EPS = 1.e-14;                       // :123
constexpr int H3_NMIN_PLM = 2, H3_NMIN_PPM = 3, H3_NMIN_PQM = 4;   // :126-129
constexpr int H3_EB_MAX_PPM = 4, H3_EB_MAX_PQM = 6;     // :132-134
constexpr int H3_LD = 6;                                 // leading dimension of the boundary LU matrices
constexpr double H3_HPLIM_IH4 = 5.e-7, H3_HPLIM_IH6 = 1.e-7;       // :140-142
H3HD double h3_hplim_eb(int ord) {                       // :143-145
  return ord <= 4 ? 1.e-10 : (ord == 5 ? 1.e-8 : 1.e-7);
}

struct H3Grid {
  int nc, n_src, method, left_bndr_ord, right_bndr_ord, p_ord, ncoef;   // ncoef = p_ord + 2 rows of tdecoeff
  double *xin;                   // (n_src+1) caller's edges, kept for the merge weights
  double *x_eps;                 // 1
  double *x_edge, *h, *hi, *hci, *w;   // x_edge_src (n_src+1), h_src, hi_src, hci_src, src_dst_weight (n_src)
  double *tde, *tds;             // tdecoeff / tdscoeff (ncoef, n_src)
  double *lblu, *rblu;           // (H3_LD, H3_LD)
  int *sdi;                      // src_dst_index (n_src)
  int *n_act, *m_act, *lb_act, *rb_act, *prepared, *err;
  int *prev, *next;              // work: doubly linked list of the merge passes (n_src)
  const int *active;             // optional (library-internal callers): columns with 0 here are left out of every launch
};
struct H3Src {
  int limiting, pc_left, pc_right;
  double *u, *uel, *uer, *usl, *usr;   // u_src, edge values, edge slopes (n_src)
  double *pc;                    // polycoeff (p_ord+1, n_src)
  double *u_range, *u_eps, *uu_eps;
  int *reconstructed, *err;
  double *wk;                    // work: 3 * (n_src + 1)
};
struct H3Map {
  int n_dst;
  double *lim, *wgt;             // seg_int_lim, seg_weight (n_src + n_dst)
  int *nseg, *sdst;              // n_src_seg (n_src), seg_dst_index (n_src + n_dst)
  int *prepared, *err;
  double *hdst;                  // work (n_dst)
};

// element k (1-based, as in the reference) of a [level][column] array
#define H3A(p, k) (p)[(size_t)((k) - 1) * nc + col]
#define H3A2(p, r, k, nr) (p)[(size_t)(((k) - 1) * (nr) + ((r) - 1)) * nc + col]

H3HD double h3_max(double a, double b) { return a > b ? a : b; }
H3HD double h3_min(double a, double b) { return a < b ? a : b; }
H3HD double h3_abs(double a) { return __builtin_fabs(a); }
H3HD double h3_sign(double a, double b) { return __builtin_copysign(a, b); }
// x**n with a run-time integer n: the compiler-rt __powidf2 sequence the reference build calls
H3HD double h3_powi(double a, int b) {
  double r = 1.0;
  while (true) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return r;
}

// ---- small dense LU (mod_hor3map.F90:577-629); a(i,j) at a[(i-1) + ld*(j-1)] -------------------------------
H3HD void h3_lu_decompose(int n, double *a, int ld) {
  for (int k = 1; k <= n - 1; ++k) {
    const double q = 1.0 / a[(k - 1) + ld * (k - 1)];
    for (int i = k + 1; i <= n; ++i) {
      a[(i - 1) + ld * (k - 1)] = a[(i - 1) + ld * (k - 1)] * q;
      for (int j = k + 1; j <= n; ++j)
        a[(i - 1) + ld * (j - 1)] = a[(i - 1) + ld * (j - 1)] - a[(i - 1) + ld * (k - 1)] * a[(k - 1) + ld * (j - 1)];
    }
  }
}
H3HD void h3_lu_solve(int n, const double *lu, int ld, double *x) {
  for (int i = 2; i <= n; ++i)
    for (int j = 1; j <= i - 1; ++j) x[i - 1] = x[i - 1] - lu[(i - 1) + ld * (j - 1)] * x[j - 1];
  x[n - 1] = x[n - 1] / lu[(n - 1) + ld * (n - 1)];
  for (int i = n - 1; i >= 1; --i) {
    for (int j = i + 1; j <= n; ++j) x[i - 1] = x[i - 1] - lu[(i - 1) + ld * (j - 1)] * x[j - 1];
    x[i - 1] = x[i - 1] / lu[(i - 1) + ld * (i - 1)];
  }
}

// ---- explicit boundary edge/slope matrices (mod_hor3map.F90:913-1039); h[0..ord-1] --------------------------
H3HD void h3_edge_slope_lblu(int ord, const double *h, double *a) {
  const int ld = H3_LD;
#define AA(i, j) a[((i) - 1) + ld * ((j) - 1)]
  double a2sq[H3_LD + 1], hsq[H3_LD + 1];
  for (int i = 1; i <= ord; ++i) AA(i, 1) = 1.0;
  AA(1, 2) = 0.5 * h[0];
  for (int i = 2; i <= ord; ++i) AA(i, 2) = AA(i - 1, 2) + 0.5 * (h[i - 2] + h[i - 1]);
  if (ord > 2) {
    AA(1, 3) = (1.0 / 3.0) * AA(1, 2) * h[0];
    for (int i = 2; i <= ord; ++i) {
      a2sq[i] = AA(i, 2) * AA(i, 2);
      hsq[i] = h[i - 1] * h[i - 1];
      AA(i, 3) = 0.5 * (a2sq[i] + (1.0 / 12.0) * hsq[i]);
    }
    if (ord > 3) {
      AA(1, 4) = 0.25 * AA(1, 3) * h[0];
      for (int i = 2; i <= ord; ++i) AA(i, 4) = (1.0 / 6.0) * AA(i, 2) * (a2sq[i] + 0.25 * hsq[i]);
      if (ord > 4) {
        AA(1, 5) = (1.0 / 5.0) * AA(1, 4) * h[0];
        for (int i = 2; i <= ord; ++i)
          AA(i, 5) = (1.0 / 24.0) * (a2sq[i] * (a2sq[i] + 0.5 * hsq[i]) + (1.0 / 80.0) * hsq[i] * hsq[i]);
        if (ord > 5) {
          AA(1, 6) = (1.0 / 6.0) * AA(1, 5) * h[0];
          for (int i = 2; i <= ord; ++i)
            AA(i, 6) = (1.0 / 120.0) * AA(i, 2) * (a2sq[i] + 0.75 * hsq[i]) * (a2sq[i] + (1.0 / 12.0) * hsq[i]);
        }
      }
    }
  }
  h3_lu_decompose(ord, a, ld);
}
H3HD void h3_edge_slope_rblu(int ord, const double *h, double *a) {
  const int ld = H3_LD;
  double a2sq[H3_LD + 1], hsq[H3_LD + 1];
  for (int i = 1; i <= ord; ++i) AA(i, 1) = 1.0;
  AA(ord, 2) = -0.5 * h[ord - 1];
  for (int i = ord - 1; i >= 1; --i) AA(i, 2) = AA(i + 1, 2) - 0.5 * (h[i] + h[i - 1]);
  if (ord > 2) {
    for (int i = 1; i <= ord - 1; ++i) {
      a2sq[i] = AA(i, 2) * AA(i, 2);
      hsq[i] = h[i - 1] * h[i - 1];
      AA(i, 3) = 0.5 * (a2sq[i] + (1.0 / 12.0) * hsq[i]);
    }
    AA(ord, 3) = -(1.0 / 3.0) * AA(ord, 2) * h[ord - 1];
    if (ord > 3) {
      for (int i = 1; i <= ord - 1; ++i) AA(i, 4) = (1.0 / 6.0) * AA(i, 2) * (a2sq[i] + 0.25 * hsq[i]);
      AA(ord, 4) = -0.25 * AA(ord, 3) * h[ord - 1];
      if (ord > 4) {
        for (int i = 1; i <= ord - 1; ++i)
          AA(i, 5) = (1.0 / 24.0) * (a2sq[i] * (a2sq[i] + 0.5 * hsq[i]) + (1.0 / 80.0) * hsq[i] * hsq[i]);
        AA(ord, 5) = -(1.0 / 5.0) * AA(ord, 4) * h[ord - 1];
        if (ord > 5) {
          for (int i = 1; i <= ord - 1; ++i)
            AA(i, 6) = (1.0 / 120.0) * AA(i, 2) * (a2sq[i] + 0.75 * hsq[i]) * (a2sq[i] + (1.0 / 12.0) * hsq[i]);
          AA(ord, 6) = -(1.0 / 6.0) * AA(ord, 5) * h[ord - 1];
        }
      }
    }
  }
  h3_lu_decompose(ord, a, ld);
#undef AA
}

// The same with the order a compile-time constant: every index is then a constant and the 6 x 6 work matrix lives in registers
// instead of scratch memory (what the run-time order costs: 300-350 bytes of scratch per lane in prepare and reconstruct).
template <int N> H3HD void h3_bndr_lu_store(const H3Grid &g, int col, bool left, int j0) {
  const int nc = g.nc;
  double hb[N], a[H3_LD * H3_LD];
#pragma unroll
  for (int i = 1; i <= N; ++i) hb[i - 1] = H3A(g.h, j0 + i);
  if (left) h3_edge_slope_lblu(N, hb, a);
  else h3_edge_slope_rblu(N, hb, a);
  double *dst = left ? g.lblu : g.rblu;
#pragma unroll
  for (int j = 1; j <= N; ++j)
#pragma unroll
    for (int i = 1; i <= N; ++i) H3A2(dst, i, j, H3_LD) = a[(i - 1) + H3_LD * (j - 1)];
}
H3HD void h3_bndr_lu(const H3Grid &g, int col, bool left, int ord, int j0) {     // cells j0+1 .. j0+ord
  switch (ord) {
    case 2: h3_bndr_lu_store<2>(g, col, left, j0); break;
    case 3: h3_bndr_lu_store<3>(g, col, left, j0); break;
    case 4: h3_bndr_lu_store<4>(g, col, left, j0); break;
    case 5: h3_bndr_lu_store<5>(g, col, left, j0); break;
    default: h3_bndr_lu_store<6>(g, col, left, j0); break;
  }
}
// first unknown of the boundary system (the edge value): lu from the planes, right-hand side u(j0+1 .. j0+N)
template <int N> H3HD double h3_bndr_solve_first(const double *lup, const double *u, size_t nc, int col, int j0) {
  double x[N], lu[H3_LD * H3_LD];
#pragma unroll
  for (int i = 1; i <= N; ++i) x[i - 1] = H3A(u, j0 + i);
#pragma unroll
  for (int j = 1; j <= N; ++j)
#pragma unroll
    for (int i = 1; i <= N; ++i) lu[(i - 1) + H3_LD * (j - 1)] = H3A2(lup, i, j, H3_LD);
  h3_lu_solve(N, lu, H3_LD, x);
  return x[0];
}
H3HD double h3_bndr_first(const double *lup, const double *u, size_t nc, int col, int ord, int j0) {
  switch (ord) {
    case 2: return h3_bndr_solve_first<2>(lup, u, nc, col, j0);
    case 3: return h3_bndr_solve_first<3>(lup, u, nc, col, j0);
    case 4: return h3_bndr_solve_first<4>(lup, u, nc, col, j0);
    case 5: return h3_bndr_solve_first<5>(lup, u, nc, col, j0);
    default: return h3_bndr_solve_first<6>(lup, u, nc, col, j0);
  }
}

// ---- merging of thin cells at the boundaries (mod_hor3map.F90:431-575) ---------------------------------------
H3HD void h3_left_bndr_cond(const H3Grid &g, int col, int &first_index, int &last_index, int &lb_ord, int &ns,
                            int ns_min) {
  const int nc = g.nc;
  int jf = first_index;
  while (true) {
    int j = jf;
    double hp = H3A(g.h, j), h_max = H3A(g.h, j);
    for (int n = 1; n <= lb_ord - 1; ++n) {
      j = H3A(g.next, j);
      hp = hp * H3A(g.h, j);
      h_max = h3_max(h_max, H3A(g.h, j));
    }
    if (hp > h3_hplim_eb(lb_ord) * h3_powi(h_max, lb_ord)) return;
    ns = ns - 1;
    if (ns < ns_min) return;
    j = jf;
    double h_min = H3A(g.h, j);
    int j_min = j;
    for (int n = 1; n <= lb_ord - 1; ++n) {
      j = H3A(g.next, j);
      if (H3A(g.h, j) < h_min) { h_min = H3A(g.h, j); j_min = j; }
    }
    const int jp = H3A(g.prev, j_min), jn = H3A(g.next, j_min);
    if (jp == 0) {
      H3A(g.sdi, j_min) = -jn;
      H3A(g.h, jn) = H3A(g.h, jn) + H3A(g.h, j_min);
      first_index = jn;
      H3A(g.prev, jn) = 0;
      jf = jn;
    } else if (jn == 0) {
      H3A(g.sdi, j_min) = -jp;
      H3A(g.h, jp) = H3A(g.h, jp) + H3A(g.h, j_min);
      H3A(g.next, jp) = 0;
      last_index = jp;
    } else {
      if (H3A(g.h, jn) < H3A(g.h, jp)) {
        H3A(g.sdi, j_min) = -jn;
        H3A(g.h, jn) = H3A(g.h, jn) + H3A(g.h, j_min);
      } else {
        H3A(g.sdi, j_min) = -jp;
        H3A(g.h, jp) = H3A(g.h, jp) + H3A(g.h, j_min);
      }
      H3A(g.next, jp) = jn;
      H3A(g.prev, jn) = jp;
    }
    lb_ord = ns < lb_ord ? ns : lb_ord;
  }
}
H3HD void h3_right_bndr_cond(const H3Grid &g, int col, int last_index, int &rb_ord, int &ns, int ns_min) {
  const int nc = g.nc;
  int jl = last_index;
  while (true) {
    int j = jl;
    double hp = H3A(g.h, j), h_max = H3A(g.h, j);
    for (int n = 1; n <= rb_ord - 1; ++n) {
      j = H3A(g.prev, j);
      hp = hp * H3A(g.h, j);
      h_max = h3_max(h_max, H3A(g.h, j));
    }
    if (hp > h3_hplim_eb(rb_ord) * h3_powi(h_max, rb_ord)) return;
    ns = ns - 1;
    if (ns < ns_min) return;
    j = jl;
    double h_min = H3A(g.h, j);
    int j_min = j;
    for (int n = 1; n <= rb_ord - 1; ++n) {
      j = H3A(g.prev, j);
      if (H3A(g.h, j) < h_min) { h_min = H3A(g.h, j); j_min = j; }
    }
    const int jp = H3A(g.prev, j_min), jn = H3A(g.next, j_min);
    if (jp == 0) {
      H3A(g.sdi, j_min) = -jn;
      H3A(g.h, jn) = H3A(g.h, jn) + H3A(g.h, j_min);
      H3A(g.prev, jn) = 0;
    } else if (jn == 0) {
      H3A(g.sdi, j_min) = -jp;
      H3A(g.h, jp) = H3A(g.h, jp) + H3A(g.h, j_min);
      H3A(g.next, jp) = 0;
      jl = jp;
    } else {
      if (H3A(g.h, jn) < H3A(g.h, jp)) {
        H3A(g.sdi, j_min) = -jn;
        H3A(g.h, jn) = H3A(g.h, jn) + H3A(g.h, j_min);
      } else {
        H3A(g.sdi, j_min) = -jp;
        H3A(g.h, jp) = H3A(g.h, jp) + H3A(g.h, j_min);
      }
      H3A(g.next, jp) = jn;
      H3A(g.prev, jn) = jp;
    }
    rb_ord = ns < rb_ord ? ns : rb_ord;
  }
}

// continuous array of the kept cells' edges (mod_hor3map.F90:1461-1472 and the PLM/PCM twins)
H3HD void h3_continuous_edges(const H3Grid &g, int col, int ns) {
  const int nc = g.nc;
  H3A(g.x_edge, 1) = H3A(g.xin, 1);
  // the reference scans, for every kept cell j, forward to the next source index whose destination is neither j nor none; every
  // source index is looked at once, in order, so the scan runs over the source indices with eight levels' loads in flight
  int j = 1;
  for (int js0 = 2; js0 <= g.n_src && j <= ns - 1; js0 += 8) {
    int a_d[8];
    double a_x[8];
    for (int q = 0; q < 8; ++q) {
      const int jq = js0 + q <= g.n_src ? js0 + q : g.n_src;
      a_d[q] = H3A(g.sdi, jq); a_x[q] = H3A(g.xin, jq);
    }
    for (int q = 0; q < 8; ++q) {
      if (js0 + q > g.n_src || j > ns - 1) break;
      if (a_d[q] != j && a_d[q] != 0) {
        H3A(g.x_edge, j + 1) = a_x[q];
        j = j + 1;
      }
    }
  }
  H3A(g.x_edge, ns + 1) = H3A(g.xin, g.n_src + 1);
}

// resolve merge chains to destination indices and weights, compact widths (mod_hor3map.F90:1430-1459)
H3HD void h3_compact_and_weights(const H3Grid &g, int col) {
  const int nc = g.nc;
  // eight levels' loads in flight in both loops: a level's inputs are not written by the levels before it (the compacted index
  // never runs ahead of the source index; a merge chain ends at the same cell whether or not a link on the way is resolved yet)
  int jd = 0;
  for (int js0 = 1; js0 <= g.n_src; js0 += 8) {
    int a_d[8];
    double a_h[8];
    for (int q = 0; q < 8; ++q) {
      const int jq = js0 + q <= g.n_src ? js0 + q : g.n_src;
      a_d[q] = H3A(g.sdi, jq); a_h[q] = H3A(g.h, jq);
    }
    for (int q = 0; q < 8; ++q) {
      const int js = js0 + q;
      if (js > g.n_src) break;
      if (a_d[q] > 0) {
        jd = jd + 1;
        H3A(g.sdi, js) = jd;
        H3A(g.h, jd) = a_h[q];
        H3A(g.hi, jd) = 1.0 / a_h[q];
      }
    }
  }
  const double x_eps = g.x_eps[col];
  for (int js0 = 1; js0 <= g.n_src; js0 += 8) {
    int a_d[8];
    double a_x[9], a_h[8], a_hi[8];
    for (int q = 0; q < 8; ++q) a_d[q] = H3A(g.sdi, js0 + q <= g.n_src ? js0 + q : g.n_src);
    for (int q = 0; q < 9; ++q) a_x[q] = H3A(g.xin, js0 + q <= g.n_src + 1 ? js0 + q : g.n_src + 1);
    for (int q = 0; q < 8; ++q) {
      while (a_d[q] < 0) a_d[q] = H3A(g.sdi, -a_d[q]);
      const int jq = a_d[q] > 0 ? a_d[q] : 1;
      a_h[q] = H3A(g.h, jq); a_hi[q] = H3A(g.hi, jq);
    }
    for (int q = 0; q < 8; ++q) {
      const int js = js0 + q;
      if (js > g.n_src) break;
      jd = a_d[q];
      H3A(g.sdi, js) = jd;
      if (jd > 0) {
        const double h = h3_abs(a_x[q + 1] - a_x[q]);
        if (h3_abs(h - a_h[q]) < x_eps) H3A(g.w, js) = 1.0;
        else H3A(g.w, js) = h * a_hi[q];
      }
    }
  }
}

// first pass shared by PPM and PQM: drop near-empty cells, build the linked list (mod_hor3map.F90:1323-1344)
H3HD int h3_link_nonempty(const H3Grid &g, int col, int &first_index, int &last_index) {
  const int nc = g.nc;
  const double x_eps = g.x_eps[col];
  int ns = 0, jp = 0;
  first_index = 0;
  for (int j0 = 1; j0 <= g.n_src; j0 += 8) {
    double a_x[9];
    for (int q = 0; q < 9; ++q) a_x[q] = H3A(g.xin, j0 + q <= g.n_src + 1 ? j0 + q : g.n_src + 1);
    for (int q = 0; q < 8; ++q) {
      const int j = j0 + q;
      if (j > g.n_src) break;
      const double h = h3_abs(a_x[q + 1] - a_x[q]);
      H3A(g.h, j) = h;
      if (h > 2.0 * x_eps) {
        ns = ns + 1;
        H3A(g.sdi, j) = 1;
        H3A(g.prev, j) = jp;
        if (jp == 0) first_index = j;
        else H3A(g.next, jp) = j;
        jp = j;
      } else {
        H3A(g.sdi, j) = 0;
      }
    }
  }
  last_index = jp;
  if (jp > 0) H3A(g.next, jp) = 0;
  return ns;
}

// mod_hor3map.F90:1308-1497
H3HD void h3_prepare_ppm(const H3Grid &g, int col) {
  const int nc = g.nc;
  int first_index, last_index;
  int ns = h3_link_nonempty(g, col, first_index, last_index);
  if (ns < H3_NMIN_PPM) { g.n_act[col] = ns; return; }

  // merge neighbours whose width ratio would ill-condition edge_ih4_coeff
  int jf = first_index, jl = H3A(g.next, jf);
  while (true) {
    const double hf = H3A(g.h, jf), hl = H3A(g.h, jl);
    const double hm = h3_max(hf, hl);
    if (hf * hl > H3_HPLIM_IH4 * (hm * hm)) {
      jf = jl;
      jl = H3A(g.next, jf);
      if (jl == 0) break;
    } else {
      ns = ns - 1;
      if (ns < H3_NMIN_PPM) { g.n_act[col] = ns; return; }
      if (hf < hl) {
        const int j = jf;
        jf = H3A(g.prev, jf);
        H3A(g.prev, jl) = jf;
        if (jf == 0) {
          H3A(g.sdi, j) = -jl;
          H3A(g.h, jl) = H3A(g.h, jl) + H3A(g.h, j);
          first_index = jl;
          jf = jl;
          jl = H3A(g.next, jf);
          if (jl == 0) break;
        } else {
          if (H3A(g.h, jf) < H3A(g.h, jl)) {
            H3A(g.sdi, j) = -jf;
            H3A(g.h, jf) = H3A(g.h, jf) + H3A(g.h, j);
          } else {
            H3A(g.sdi, j) = -jl;
            H3A(g.h, jl) = H3A(g.h, jl) + H3A(g.h, j);
          }
          H3A(g.next, jf) = jl;
        }
      } else {
        const int j = jl;
        jl = H3A(g.next, jl);
        H3A(g.next, jf) = jl;
        if (jl == 0) {
          H3A(g.sdi, j) = -jf;
          H3A(g.h, jf) = H3A(g.h, jf) + H3A(g.h, j);
          last_index = jf;
          break;
        }
        if (H3A(g.h, jf) < H3A(g.h, jl)) {
          H3A(g.sdi, j) = -jf;
          H3A(g.h, jf) = H3A(g.h, jf) + H3A(g.h, j);
        } else {
          H3A(g.sdi, j) = -jl;
          H3A(g.h, jl) = H3A(g.h, jl) + H3A(g.h, j);
        }
        H3A(g.prev, jl) = jf;
      }
    }
  }

  int lb_ord = ns;
  if (g.left_bndr_ord < lb_ord) lb_ord = g.left_bndr_ord;
  if (H3_EB_MAX_PPM < lb_ord) lb_ord = H3_EB_MAX_PPM;
  h3_left_bndr_cond(g, col, first_index, last_index, lb_ord, ns, H3_NMIN_PPM);
  if (ns < H3_NMIN_PPM) { g.n_act[col] = ns; return; }
  int rb_ord = ns;
  if (g.right_bndr_ord < rb_ord) rb_ord = g.right_bndr_ord;
  if (H3_EB_MAX_PPM < rb_ord) rb_ord = H3_EB_MAX_PPM;
  h3_right_bndr_cond(g, col, last_index, rb_ord, ns, H3_NMIN_PPM);
  if (ns < H3_NMIN_PPM) { g.n_act[col] = ns; return; }

  h3_compact_and_weights(g, col);
  h3_continuous_edges(g, col, ns);

  // The reference stores hci_src (:1477) and the edge system's coefficients tdecoeff (edge_ih4_coeff, :631-648) here.  Their only
  // reader, the PPM reconstruction, recomputes both from the widths with the same expressions (hor3map_ppm_fused.h: the same bits),
  // so the five planes are not written: 225 of the 460 MB this routine stored per 106 k-column slab.

  if (lb_ord > 1) h3_bndr_lu(g, col, true, lb_ord, 0);
  if (rb_ord > 1) h3_bndr_lu(g, col, false, rb_ord, ns - rb_ord);
  g.n_act[col] = ns;
  g.lb_act[col] = lb_ord;
  g.rb_act[col] = rb_ord;
}

// mod_hor3map.F90:1499-1604 (PLM when with_hci, PCM otherwise)
H3HD void h3_prepare_plm_pcm(const H3Grid &g, int col, bool plm) {
  const int nc = g.nc;
  const double x_eps = g.x_eps[col];
  int ns = 0;
  for (int j = 1; j <= g.n_src; ++j) {
    if (h3_abs(H3A(g.xin, j + 1) - H3A(g.xin, j)) > 2.0 * x_eps) {
      ns = ns + 1;
      H3A(g.sdi, j) = ns;
    } else {
      H3A(g.sdi, j) = 0;
    }
  }
  if (plm ? ns < H3_NMIN_PLM : ns == 0) { g.n_act[col] = ns; return; }
  h3_continuous_edges(g, col, ns);
  for (int j = 1; j <= ns; ++j) {
    H3A(g.h, j) = h3_abs(H3A(g.x_edge, j + 1) - H3A(g.x_edge, j));
    H3A(g.hi, j) = 1.0 / H3A(g.h, j);
  }
  if (plm)
    for (int j = 2; j <= ns - 1; ++j)
      H3A(g.hci, j) = 2.0 / (H3A(g.h, j - 1) + 2.0 * H3A(g.h, j) + H3A(g.h, j + 1));
  g.n_act[col] = ns;
}

H3HD void h3_prepare_pqm(const H3Grid &g, int col);   // hor3map_pqm.h

// prepare_reconstruction (mod_hor3map.F90:3834-3945); the caller's edges are in g.xin
H3HD int h3_prepare_reconstruction(const H3Grid &g, int col) {
  const int nc = g.nc;
  const int n = g.n_src;
  if (H3A(g.xin, n + 1) - H3A(g.xin, 1) > 0.0) {
    for (int j = 1; j <= n; ++j)
      if (H3A(g.xin, j + 1) < H3A(g.xin, j)) return H3_NONMONOTONIC_SRC_EDGES;
  } else {
    for (int j = 1; j <= n; ++j)
      if (H3A(g.xin, j + 1) > H3A(g.xin, j)) return H3_NONMONOTONIC_SRC_EDGES;
  }
  g.prepared[col] = 0;   // only past the monotonicity check, as in the reference (:3898)
  g.x_eps[col] = h3_max(h3_abs(H3A(g.xin, n + 1) - H3A(g.xin, 1)), H3_EPS) * H3_EPS;
  int m = g.method;
  if (m == H3_PQM) {
    h3_prepare_pqm(g, col);
    if (g.n_act[col] < H3_NMIN_PQM) m = H3_PPM;
  }
  if (m == H3_PPM) {
    h3_prepare_ppm(g, col);
    if (g.n_act[col] < H3_NMIN_PPM) m = H3_PLM;
  }
  if (m == H3_PLM) {
    h3_prepare_plm_pcm(g, col, true);
    if (g.n_act[col] < H3_NMIN_PLM) m = H3_PCM;
  }
  g.m_act[col] = m;
  if (m == H3_PCM) {
    h3_prepare_plm_pcm(g, col, false);
    if (g.n_act[col] == 0) return H3_SRC_EXTENT_TOO_SMALL;
  }
  g.prepared[col] = 1;
  return H3_NOERR;
}

// ---- reconstruction ------------------------------------------------------------------------------------------
#define PC(c, j) H3A2(s.pc, c, j, np)

// mod_hor3map.F90:1606-1705
H3HD void h3_reconstruct_plm(const H3Grid &g, const H3Src &s, int col, bool limited) {
  const int nc = g.nc, np = g.p_ord + 1;
  const int ns = g.n_act[col];
  double sc;
  for (int j = 2; j <= ns - 1; ++j) {
    if (limited) {
      const double sl = 2.0 * (H3A(s.u, j) - H3A(s.u, j - 1)) * H3A(g.hi, j);
      const double sr = 2.0 * (H3A(s.u, j + 1) - H3A(s.u, j)) * H3A(g.hi, j);
      if (sl * sr > 0.0) {
        sc = (H3A(s.u, j + 1) - H3A(s.u, j - 1)) * H3A(g.hci, j);
        sc = h3_sign(h3_min(h3_min(h3_abs(sl), h3_abs(sr)), h3_abs(sc)), sc);
      } else {
        sc = 0.0;
      }
    } else {
      sc = (H3A(s.u, j + 1) - H3A(s.u, j - 1)) * H3A(g.hci, j);
    }
    PC(2, j) = sc * H3A(g.h, j);
    PC(1, j) = H3A(s.u, j) - 0.5 * PC(2, j);
    H3A(s.uel, j) = PC(1, j);
    H3A(s.uer, j) = PC(1, j) + PC(2, j);
  }
  if (limited && s.pc_left) {
    PC(1, 1) = H3A(s.u, 1);
    PC(2, 1) = 0.0;
    H3A(s.uel, 1) = H3A(s.u, 1);
    H3A(s.uer, 1) = H3A(s.u, 1);
  } else {
    sc = 2.0 * (H3A(s.u, 2) - H3A(s.u, 1)) / (H3A(g.h, 2) + H3A(g.h, 1));
    PC(2, 1) = sc * H3A(g.h, 1);
    PC(1, 1) = H3A(s.u, 1) - 0.5 * PC(2, 1);
    H3A(s.uel, 1) = PC(1, 1);
    H3A(s.uer, 1) = PC(1, 1) + PC(2, 1);
  }
  if (limited && s.pc_right) {
    PC(1, ns) = H3A(s.u, ns);
    PC(2, ns) = 0.0;
    H3A(s.uel, ns) = H3A(s.u, ns);
    H3A(s.uer, ns) = H3A(s.u, ns);
  } else {
    sc = 2.0 * (H3A(s.u, ns) - H3A(s.u, ns - 1)) / (H3A(g.h, ns) + H3A(g.h, ns - 1));
    PC(2, ns) = sc * H3A(g.h, ns);
    PC(1, ns) = H3A(s.u, ns) - 0.5 * PC(2, ns);
    H3A(s.uel, ns) = PC(1, ns);
    H3A(s.uer, ns) = PC(1, ns) + PC(2, ns);
  }
  // polycoeff(3:p_ord+1,:) = 0 over the whole allocated extent
  for (int c = 3; c <= np; ++c)
    for (int j = 1; j <= g.n_src; ++j) PC(c, j) = 0.0;
}

H3HD void h3_ppm_edge_consistency(const H3Src &s, int nc, int col, int ns) {   // :1908-1914
  for (int j = 3; j <= ns - 1; ++j)
    if ((H3A(s.uel, j) - H3A(s.uer, j - 1)) * (H3A(s.u, j) - H3A(s.u, j - 1)) < 0.0) {
      H3A(s.uel, j) = 0.5 * (H3A(s.uer, j - 1) + H3A(s.uel, j));
      H3A(s.uer, j - 1) = H3A(s.uel, j);
    }
}
H3HD void h3_reconstruct_pqm(const H3Grid &g, const H3Src &s, int col);   // hor3map_pqm.h
H3HD void h3_reconstruct_ppm_fused(const H3Grid &g, const H3Src &s, int col);   // hor3map_ppm_fused.h

// reconstruct (mod_hor3map.F90:4145-4272); the caller's data are in uin [level][column]
H3HD int h3_reconstruct(const H3Grid &g, const H3Src &s, const double *uin, int col) {
  const int nc = g.nc;
  if (!g.prepared[col]) return H3_RECON_NOT_PREPARED;
  const int m = g.m_act[col], ns = g.n_act[col];
  if (m == H3_PCM || m == H3_PLM) {
    for (int js = 1; js <= g.n_src; ++js) {
      const int jd = H3A(g.sdi, js);
      if (jd != 0) H3A(s.u, jd) = H3A(uin, js);
    }
  } else {
    // The reference zeroes u_src and accumulates w * u into memory (:4197-4204).  The source cells of a destination cell are
    // neighbours, so its sum lives in a register from 0.0 on, with the same additions in the same order, and is stored once; a
    // destination index that comes back after another one (no merge rule produces it) picks its partial sum up from memory.
    int cur = 0, top = 0;
    double acc = 0.0;
    for (int js0 = 1; js0 <= g.n_src; js0 += 8) {                      // eight levels' loads in flight (they do not depend on the sums)
      int a_d[8];
      double a_w[8], a_u[8];
      for (int q = 0; q < 8; ++q) {
        const int jq = js0 + q <= g.n_src ? js0 + q : g.n_src;
        a_d[q] = H3A(g.sdi, jq); a_w[q] = H3A(g.w, jq); a_u[q] = H3A(uin, jq);
      }
      for (int q = 0; q < 8; ++q) {
        if (js0 + q > g.n_src) break;
        const int jd = a_d[q];
        if (jd == 0) continue;
        if (jd != cur) {
          if (cur > 0) H3A(s.u, cur) = acc;
          for (int z = top + 1; z < jd; ++z) H3A(s.u, z) = 0.0;        // cells no source maps to keep the reference's zero
          acc = jd > top ? 0.0 : H3A(s.u, jd);
          cur = jd;
          if (jd > top) top = jd;
        }
        acc = acc + a_w[q] * a_u[q];
      }
    }
    if (cur > 0) H3A(s.u, cur) = acc;
    for (int q = top + 1; q <= ns; ++q) H3A(s.u, q) = 0.0;
  }
  double umin = H3A(s.u, 1), umax = umin;
  for (int j0 = 2; j0 <= ns; j0 += 8) {
    double a_u[8];
    for (int q = 0; q < 8; ++q) a_u[q] = H3A(s.u, j0 + q <= ns ? j0 + q : ns);
    for (int q = 0; q < 8; ++q) {
      if (j0 + q > ns) break;
      umin = h3_min(umin, a_u[q]);
      umax = h3_max(umax, a_u[q]);
    }
  }
  const double u_range = h3_abs(umin - umax);
  s.u_range[col] = u_range;
  const double u_eps = h3_max(u_range, H3_EPS * H3_EPS) * H3_EPS;
  s.u_eps[col] = u_eps;
  s.uu_eps[col] = h3_max(u_range, H3_EPS * H3_EPS) * u_eps;
  const int lim = s.limiting;
  const bool known = lim == H3_NO_LIMITING || lim == H3_MONOTONIC || lim == H3_NON_OSCILLATORY ||
                     lim == H3_NON_OSCILLATORY_POSDEF;
  if (m == H3_PLM) {
    if (!known) return H3_INVALID_PLM_LIMITING;
    h3_reconstruct_plm(g, s, col, lim != H3_NO_LIMITING);
  } else if (m == H3_PPM) {
    if (!known) return H3_INVALID_PPM_LIMITING;
    h3_reconstruct_ppm_fused(g, s, col);        // hor3map_ppm_fused.h
  } else if (m == H3_PQM) {
    if (!known) return H3_INVALID_PQM_LIMITING;
    h3_reconstruct_pqm(g, s, col);
  }
  s.reconstructed[col] = 1;
  return H3_NOERR;
}

// extract_polycoeff (mod_hor3map.F90:4274-4459); out is (p_ord+1, n_src) in [coef+np*(cell)][column]
H3HD int h3_extract_polycoeff(const H3Grid &g, const H3Src &s, double *out, int col) {
  const int nc = g.nc, np = g.p_ord + 1, n = g.n_src;
  if (!s.reconstructed[col]) return H3_RECON_NOT_AVAILABLE;
#define OUT(c, j) H3A2(out, c, j, np)
  // the reference zeroes polycoeff first (:4290); a column on the grid's own PPM / PQM path writes every element below exactly
  // once instead (zeros included), only a column that fell back to a lower order is pre-filled
  const int m = g.m_act[col];
  const bool full = (m == H3_PPM && np == 3) || (m == H3_PQM && np == 5);
  if (!full)
    for (int j = 1; j <= n; ++j)
      for (int c = 1; c <= np; ++c) OUT(c, j) = 0.0;
  int js0 = 1, jd;
  if (m == H3_PCM) {
    while (true) {
      jd = H3A(g.sdi, js0);
      if (jd == 0) OUT(1, js0) = H3A(s.u, 1);
      else { OUT(1, js0) = H3A(s.u, jd); break; }
      js0 = js0 + 1;
      if (js0 > n) break;
    }
    for (int js = js0 + 1; js <= n; ++js) {
      jd = H3A(g.sdi, js);
      OUT(1, js) = jd == 0 ? OUT(1, js - 1) : H3A(s.u, jd);
    }
  } else if (m == H3_PLM) {
    while (true) {
      jd = H3A(g.sdi, js0);
      if (jd == 0) OUT(1, js0) = PC(1, 1);
      else { OUT(1, js0) = PC(1, 1); OUT(2, js0) = PC(2, 1); break; }
      js0 = js0 + 1;
      if (js0 > n) break;
    }
    for (int js = js0 + 1; js <= n; ++js) {
      jd = H3A(g.sdi, js);
      if (jd == 0) OUT(1, js) = OUT(1, js - 1) + OUT(2, js - 1);
      else { OUT(1, js) = PC(1, jd); OUT(2, js) = PC(2, jd); }
    }
  } else {
    const int nq = m == H3_PPM ? 3 : 5;
    while (true) {
      jd = H3A(g.sdi, js0);
      if (jd == 0) {
        OUT(1, js0) = PC(1, 1);
        if (full) for (int c = 2; c <= nq; ++c) OUT(c, js0) = 0.0;
      } else break;
      js0 = js0 + 1;
      if (js0 > n) break;
    }
    int jd_prev = -1;
    double xi0 = 0.0;
    // four levels' loads in flight (destination indices and weights first, then the coefficients they point to); the cell above,
    // which a near-empty cell continues, travels in registers instead of being read back
    double pv[5] = {js0 > 1 ? PC(1, 1) : 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int jb = js0; jb <= n; jb += 4) {
      int a_d[4];
      double a_w[4], a_p[4][5];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jq = jb + q <= n ? jb + q : n;
        a_d[q] = H3A(g.sdi, jq); a_w[q] = H3A(g.w, jq);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jq = a_d[q] > 0 ? a_d[q] : 1;
#pragma unroll
        for (int c = 1; c <= 5; ++c) a_p[q][c - 1] = c <= nq ? PC(c, jq) : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int js = jb + q;
        if (js > n) break;
        jd = a_d[q];
        double o[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        if (jd == 0) {
          double acc = pv[0] + pv[1];
#pragma unroll
          for (int c = 3; c <= 5; ++c) if (c <= nq) acc = acc + pv[c - 1];
          o[0] = acc;
          OUT(1, js) = acc;
          if (full) {
#pragma unroll
            for (int c = 2; c <= 5; ++c) if (c <= nq) OUT(c, js) = 0.0;
          }
        } else {
          const double w = a_w[q];
          const double *pc = a_p[q];
          if (w == 1.0) {
#pragma unroll
            for (int c = 1; c <= 5; ++c) if (c <= nq) o[c - 1] = pc[c - 1];
          } else {
            if (jd != jd_prev) xi0 = 0.0;
            jd_prev = jd;
            double qq = w;
            if (nq == 3) {
              o[0] = (pc[2] * xi0 + pc[1]) * xi0 + pc[0];
              o[1] = (2.0 * pc[2] * xi0 + pc[1]) * qq;
              qq = qq * w;
              o[2] = pc[2] * qq;
            } else {
              o[0] = (((pc[4] * xi0 + pc[3]) * xi0 + pc[2]) * xi0 + pc[1]) * xi0 + pc[0];
              o[1] = (((4.0 * pc[4] * xi0 + 3.0 * pc[3]) * xi0 + 2.0 * pc[2]) * xi0 + pc[1]) * qq;
              qq = qq * w;
              o[2] = ((6.0 * pc[4] * xi0 + 3.0 * pc[3]) * xi0 + pc[2]) * qq;
              qq = qq * w;
              o[3] = (4.0 * pc[4] * xi0 + pc[3]) * qq;
              qq = qq * w;
              o[4] = pc[4] * qq;
            }
            xi0 = xi0 + w;
          }
#pragma unroll
          for (int c = 1; c <= 5; ++c) if (c <= nq) OUT(c, js) = o[c - 1];
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) pv[c] = o[c];
      }
    }
  }
#undef OUT
  return H3_NOERR;
}

// ---- regridding ----------------------------------------------------------------------------------------------
// mod_hor3map.F90:2962-3027
H3HD double h3_line_intersection(double p1, double p2, double u, double u_eps, double xil, double xir) {
  if (h3_abs(p2) < u_eps) return xil;
  return h3_max(xil, h3_min(xir, (u - p1) / p2));
}
H3HD double h3_parabola_intersection(double p1, double p2, double p3, double u, double u_eps, double xil,
                                     double xir) {
  if (h3_abs(p3) < u_eps) return h3_line_intersection(p1, p2, u, u_eps, xil, xir);
  const double q = 0.5 / p3;
  const double sq = __builtin_sqrt(h3_max(0.0, p2 * p2 - 4.0 * p3 * (p1 - u)));
  const double xi1 = -(p2 + sq) * q, xi2 = -(p2 - sq) * q;
  const double xim = 0.5 * (xil + xir);
  const double xi = h3_abs(xi1 - xim) < h3_abs(xi2 - xim) ? xi1 : xi2;
  return h3_max(xil, h3_min(xir, xi));
}
H3HD double h3_quartic_intersection(const double *p, double u, double u_eps, double xil, double xir) {
  if (h3_abs(p[3]) < u_eps && h3_abs(p[4]) < u_eps)
    return h3_parabola_intersection(p[0], p[1], p[2], u, u_eps, xil, xir);
  double xi = 0.5 * (xil + xir);
  for (int n = 1; n <= 10; ++n) {
    const double r = p[0] + (p[1] + (p[2] + (p[3] + p[4] * xi) * xi) * xi) * xi - u;
    const double drdx = p[1] + (2.0 * p[2] + (3.0 * p[3] + 4.0 * p[4] * xi) * xi) * xi;
    const double xi_old = xi;
    xi = h3_max(xil, h3_min(xir, xi_old - r / h3_sign(h3_max(H3_EPS, h3_abs(drdx)), drdx)));
    if (h3_abs(xi - xi_old) < 1.e-9) return xi;
  }
  return xi;
}
// intersection with the reconstruction of cell js, any order
H3HD double h3_cell_intersection(const H3Grid &g, const H3Src &s, int col, int m, int js, double u, double xil,
                                 double xir) {
  const int nc = g.nc, np = g.p_ord + 1;
  const double u_eps = s.u_eps[col];
  if (m == H3_PLM) return h3_line_intersection(PC(1, js), PC(2, js), u, u_eps, xil, xir);
  if (m == H3_PPM) return h3_parabola_intersection(PC(1, js), PC(2, js), PC(3, js), u, u_eps, xil, xir);
  const double p[5] = {PC(1, js), PC(2, js), PC(3, js), PC(4, js), PC(5, js)};
  return h3_quartic_intersection(p, u, u_eps, xil, xir);
}
#define XEDGE(js, xi) (H3A(g.x_edge, js) + (H3A(g.x_edge, (js) + 1) - H3A(g.x_edge, js)) * (xi))

// regrid_{plm,ppm,pqm}_method_1 (mod_hor3map.F90:3029-3207)
// The walk over source cells (js) and target values (jg) with what either index will need next loaded one advance ahead: the
// current cell's edge values, edges and coefficients and the current target live in registers.  Statements and order as in the
// reference.
struct H3RgCell { double uer, uel_nx, xlo, xhi, pc[5]; };
H3HD double h3_cell_intersection_pc(int m, const double *pc, double u, double u_eps, double xil, double xir) {
  if (m == H3_PLM) return h3_line_intersection(pc[0], pc[1], u, u_eps, xil, xir);
  if (m == H3_PPM) return h3_parabola_intersection(pc[0], pc[1], pc[2], u, u_eps, xil, xir);
  return h3_quartic_intersection(pc, u, u_eps, xil, xir);
}
H3HD void h3_regrid_method_1(const H3Grid &g, const H3Src &s, int col, int m, double u_sgn, int ng,
                             const double *ugrd, double *xgrd) {
  const int nc = g.nc, np = g.p_ord + 1;
  const int ns = g.n_act[col];
  const double u_eps = s.u_eps[col];
  const int nq = m == H3_PLM ? 2 : (m == H3_PPM ? 3 : 5);
  auto load_cell = [&](H3RgCell &c, int j) {
    const int jj = j <= ns ? j : ns, jn = jj + 1 <= ns ? jj + 1 : ns;
    c.uer = H3A(s.uer, jj); c.uel_nx = H3A(s.uel, jn);
    c.xlo = H3A(g.x_edge, jj); c.xhi = H3A(g.x_edge, jj + 1);
#pragma unroll
    for (int q = 1; q <= 5; ++q) c.pc[q - 1] = q <= nq ? PC(q, jj) : 0.0;
  };
  auto ug_at = [&](int j) { return H3A(ugrd, j <= ng ? j : ng); };
  int jg = 1;
  double ug = ug_at(1), ug_nx = ug_at(2);
#define H3RG_NEXT_TARGET() { jg = jg + 1; if (jg > ng) return; ug = ug_nx; ug_nx = ug_at(jg + 1); }
  const double uel1 = H3A(s.uel, 1);
  while (true) {
    if ((ug - uel1) * u_sgn >= 0.0) break;
    H3RG_NEXT_TARGET();
  }
  int js = 1;
  H3RgCell c, cn;
  load_cell(c, 1);
  load_cell(cn, 2);
  while (true) {
    if (js + 1 > ns) break;
    const double ue_min = h3_min(c.uer * u_sgn, c.uel_nx * u_sgn);
    while (true) {
      if (ug * u_sgn >= ue_min) break;
      const double xi = h3_cell_intersection_pc(m, c.pc, ug, u_eps, 0.0, 1.0);
      H3A(xgrd, jg) = c.xlo + (c.xhi - c.xlo) * xi;
      H3RG_NEXT_TARGET();
    }
    const double ue_max = h3_max(c.uer * u_sgn, c.uel_nx * u_sgn);
    while (true) {
      if (ug * u_sgn > ue_max) break;
      H3A(xgrd, jg) = c.xhi;
      H3RG_NEXT_TARGET();
    }
    js = js + 1;
    c = cn;
    load_cell(cn, js + 1);
  }
  while (true) {
    if ((ug - c.uer) * u_sgn > 0.0) return;
    const double xi = h3_cell_intersection_pc(m, c.pc, ug, u_eps, 0.0, 1.0);
    H3A(xgrd, jg) = c.xlo + (c.xhi - c.xlo) * xi;
    H3RG_NEXT_TARGET();
  }
#undef H3RG_NEXT_TARGET
}

// value and derivative (d/dxi) of the reconstruction at the cell mid point
H3HD double h3_mid_value(const H3Src &s, int nc, int np, int col, int m, int js) {
  if (m == H3_PLM) return PC(1, js) + 0.5 * PC(2, js);
  if (m == H3_PPM) return PC(1, js) + 0.5 * PC(2, js) + 0.25 * PC(3, js);
  return PC(1, js) + 0.5 * PC(2, js) + 0.25 * PC(3, js) + 0.125 * PC(4, js) + (1.0 / 16.0) * PC(5, js);
}
H3HD double h3_mid_slope(const H3Src &s, int nc, int np, int col, int m, int js) {
  if (m == H3_PLM) return PC(2, js);
  if (m == H3_PPM) return PC(2, js) + PC(3, js);
  return PC(2, js) + PC(3, js) + 0.75 * PC(4, js) + 0.5 * PC(5, js);
}

// regrid_{plm,ppm,pqm}_method_2 (mod_hor3map.F90:3209-3605)
H3HD void h3_regrid_method_2(const H3Grid &g, const H3Src &s, int col, int m, double u_sgn, int ng,
                             const double *ugrd, double *xgrd) {
  const int nc = g.nc, np = g.p_ord + 1;
  const int ns = g.n_act[col];
  const double u_eps = s.u_eps[col];
  // the current target value lives in a register, the next one is loaded one advance ahead
  auto ug_at = [&](int j) { return H3A(ugrd, j <= ng ? j : ng); };
  int jg = 1;
  double ug = ug_at(1), ug_nx = ug_at(2);
#define H3RG2_NEXT_TARGET() { jg = jg + 1; if (jg > ng) return; ug = ug_nx; ug_nx = ug_at(jg + 1); }
  while (true) {
    if ((ug - H3A(s.uel, 1)) * u_sgn >= 0.0) break;
    H3RG2_NEXT_TARGET();
  }
  int js = 1;
  double umr = h3_mid_value(s, nc, np, col, m, js), uml;
  while (true) {
    if ((ug - umr) * u_sgn > 0.0) break;
    const double xi = h3_cell_intersection(g, s, col, m, js, ug, 0.0, 0.5);
    H3A(xgrd, jg) = XEDGE(js, xi);
    H3RG2_NEXT_TARGET();
  }
  bool done = false;
  while (!done) {
    while (true) {
      uml = umr;
      umr = h3_mid_value(s, nc, np, col, m, js);
      if ((ug - umr) * u_sgn <= 0.0) break;
      js = js + 1;
      if (js > ns) { done = true; break; }
    }
    if (done) break;
    if (js < 2) return;   // unreachable for finite data (the reference would index cell 0 here)
    const double duml = h3_mid_slope(s, nc, np, col, m, js - 1);
    const double dumr = h3_mid_slope(s, nc, np, col, m, js);
    double pcl1, pcl2, pcl3, pcr1, pcr2, pcr3;
    pcr2 = (4.0 * (umr - uml) - duml - dumr) * H3A(g.h, js) / (H3A(g.h, js - 1) + H3A(g.h, js));
    pcr1 = umr - 0.25 * (dumr + pcr2);
    if (pcr2 * (H3A(s.u, js) - H3A(s.u, js - 1)) < 0.0) {
      pcr2 = 0.0;
      const double uerl = H3A(s.uer, js - 1), uelr = H3A(s.uel, js);
      pcr1 = h3_min(h3_max(pcr1, h3_min(uerl, uelr)), h3_max(uerl, uelr));
      pcr3 = 4.0 * (umr - pcr1);
      pcl1 = 4.0 * uml - 3.0 * pcr1;
      pcl2 = 2.0 * (pcr1 - pcl1);
      pcl3 = -0.5 * pcl2;
    } else {
      pcr3 = dumr - pcr2;
      pcl1 = pcr1 - duml;
      pcl2 = 4.0 * (uml - pcl1) - duml;
      pcl3 = duml - pcl2;
    }
    while (true) {
      if ((ug - pcr1) * u_sgn > 0.0) break;
      const double xi = h3_parabola_intersection(pcl1, pcl2, pcl3, ug, u_eps, 0.5, 1.0);
      H3A(xgrd, jg) = XEDGE(js - 1, xi);
      H3RG2_NEXT_TARGET();
    }
    while (true) {
      if ((ug - umr) * u_sgn > 0.0) break;
      const double xi = h3_parabola_intersection(pcr1, pcr2, pcr3, ug, u_eps, 0.0, 0.5);
      H3A(xgrd, jg) = XEDGE(js, xi);
      H3RG2_NEXT_TARGET();
    }
  }
  js = ns;
  while (true) {
    if ((ug - H3A(s.uer, js)) * u_sgn > 0.0) return;
    const double xi = h3_cell_intersection(g, s, col, m, js, ug, 0.5, 1.0);
    H3A(xgrd, jg) = XEDGE(js, xi);
    H3RG2_NEXT_TARGET();
  }
#undef H3RG2_NEXT_TARGET
}

// regrid (mod_hor3map.F90:4461-4557)
H3HD int h3_regrid(const H3Grid &g, const H3Src &s, int col, int ng, const double *ugrd, double *xgrd,
                   double missing_value, int regrid_method) {
  const int nc = g.nc;
  if (!s.reconstructed[col]) return H3_RECON_NOT_AVAILABLE;
  for (int j = 1; j <= ng; ++j) H3A(xgrd, j) = missing_value;
  const int m = g.m_act[col];
  if (m == H3_PCM) return H3_NOERR;
  if (s.u_range[col] < H3_EPS) return H3_NOERR;
  const double u_sgn = h3_sign(1.0, H3A(s.u, g.n_act[col]) - H3A(s.u, 1));
  if (regrid_method == H3_REGRID_METHOD_1) h3_regrid_method_1(g, s, col, m, u_sgn, ng, ugrd, xgrd);
  else h3_regrid_method_2(g, s, col, m, u_sgn, ng, ugrd, xgrd);
  return H3_NOERR;
}

// ---- remapping -----------------------------------------------------------------------------------------------
// prepare_remapping (mod_hor3map.F90:3947-4143); xdst = destination edges (n_dst+1)
H3HD int h3_prepare_remapping(const H3Grid &g, const H3Map &r, const double *xdst, int col) {
  const int nc = g.nc, nd = r.n_dst;
  if (!g.prepared[col]) return H3_RECON_NOT_PREPARED;
  r.prepared[col] = 0;
  const double x_eps = g.x_eps[col];
  const int nsa = g.n_act[col];
  if (h3_abs(H3A(g.x_edge, 1) - H3A(xdst, 1)) > x_eps || h3_abs(H3A(g.x_edge, nsa + 1) - H3A(xdst, nd + 1)) > x_eps)
    return H3_INCONSISTENT_GRID_RANGE;
  const bool incr = H3A(xdst, nd + 1) - H3A(xdst, 1) > 0.0;
  if (incr) {
    for (int j = 1; j <= nd; ++j)
      if (H3A(xdst, j + 1) < H3A(xdst, j)) return H3_NONMONOTONIC_DST_EDGES;
  } else {
    for (int j = 1; j <= nd; ++j)
      if (H3A(xdst, j + 1) > H3A(xdst, j)) return H3_NONMONOTONIC_DST_EDGES;
  }
  // The reference keeps h_dst in an array and n_src_seg in memory (:4020-4031, :4046); here the width of the current destination
  // cell comes from its two edges (the same subtraction), the segment count of the current source cell lives in a register, and
  // what the walk will need when either index advances -- the next source edge with its cell's width and inverse, the next
  // destination edge -- is loaded one advance ahead, so that an iteration waits for no load issued in it.  Same statements, same order.
  const int nsrc = g.n_src;
  auto xe = [&](int q) { return H3A(g.x_edge, q <= nsrc + 1 ? q : nsrc + 1); };
  auto xdq = [&](int q) { return H3A(xdst, q <= nd + 1 ? q : nd + 1); };
  int js = 1, jd = 1;
  double xd_lo = H3A(xdst, 1), xd = H3A(xdst, 2);               // edges jd, jd+1
  double hd = h3_abs(xd - xd_lo);
  while (hd <= x_eps) {
    jd = jd + 1;
    xd_lo = xd;
    xd = xdq(jd + 1);
    hd = h3_abs(xd - xd_lo);
  }
  double xd_nx = xdq(jd + 2);                                    // edge jd+2
  double xs_lo = H3A(g.x_edge, 1), xs = xe(2), xs_nx = xe(3);   // edges js, js+1, js+2
  double hs = H3A(g.h, 1), his = H3A(g.hi, 1);
  double hs_nx = H3A(g.h, 2 <= nsrc ? 2 : nsrc), his_nx = H3A(g.hi, 2 <= nsrc ? 2 : nsrc);
  int iseg = 0, cnt = 0;
  double xil = 0.0;
  auto next_src = [&]() {                                        // js = js + 1
    H3A(r.nseg, js) = cnt;
    cnt = 0;
    js = js + 1;
    xs_lo = xs; xs = xs_nx; hs = hs_nx; his = his_nx;
    xs_nx = xe(js + 2);
    const int q = js + 1 <= nsrc ? js + 1 : nsrc;
    hs_nx = H3A(g.h, q); his_nx = H3A(g.hi, q);
  };
  auto next_dst = [&]() {                                        // jd = jd + 1
    jd = jd + 1;
    xd_lo = xd; xd = xd_nx;
    hd = h3_abs(xd - xd_lo);
    xd_nx = xdq(jd + 2);
  };
  while (true) {
    iseg = iseg + 1;
    cnt = cnt + 1;
    H3A(r.sdst, iseg) = jd;
    if (h3_abs(xs - xd) <= x_eps) {
      if (hd > x_eps) {
        H3A(r.lim, iseg) = 1.0;
        H3A(r.wgt, iseg) = (1.0 - xil) * hs / hd;
      } else {
        H3A(r.lim, iseg) = xil;
      }
      if (js == nsa) break;
      xil = 0.0;
      next_src();
      next_dst();
    } else if (incr ? xs < xd : xs > xd) {
      H3A(r.lim, iseg) = 1.0;
      H3A(r.wgt, iseg) = (1.0 - xil) * hs / hd;
      xil = 0.0;
      next_src();
    } else {
      if (hd > x_eps) {
        const double l = incr ? (xd - xs_lo) * his : (xs_lo - xd) * his;
        H3A(r.lim, iseg) = l;
        H3A(r.wgt, iseg) = (l - xil) * hs / hd;
        xil = l;
      } else {
        H3A(r.lim, iseg) = xil;
      }
      next_dst();
    }
  }
  H3A(r.nseg, js) = cnt;
  r.prepared[col] = 1;
  return H3_NOERR;
}

// remap (mod_hor3map.F90:4559-4856)
H3HD int h3_remap(const H3Grid &g, const H3Src &s, const H3Map &r, double *udst, int col) {
  const int nc = g.nc, np = g.p_ord + 1, nd = r.n_dst;
  if (!r.prepared[col]) return H3_REMAP_NOT_PREPARED;
  if (!s.reconstructed[col]) return H3_RECON_NOT_AVAILABLE;
  // The reference zeroes u_dst and accumulates segment by segment into memory; the destination index only
  // ever moves forward, so the running cell lives in a register and every destination cell is written once
  // (acc starts from 0.0 and adds in the same order: the same bits, incl. 0.0 + (-0.0) = +0.0).
  int cur = 0;
  double acc = 0.0;
#define H3_DST(jd)                                          \
  if ((jd) != cur) {                                        \
    if (cur > 0) H3A(udst, cur) = acc;                      \
    for (int q_ = cur + 1; q_ < (jd); ++q_) H3A(udst, q_) = 0.0; \
    cur = (jd);                                             \
    acc = 0.0;                                              \
  }
  const int ns = g.n_act[col], m = g.m_act[col];
  // What the walk reads is loaded ahead of its use: the source cells' segment counts, means and coefficients H3R_U cells ahead
  // (the cell index runs in lock step over the lanes), the next segment's limit, destination index and weight one segment ahead.
  // (The reference reads the weight of a segment only where it uses it; reading it always changes no result.)
#define H3R_U 4
  const int nsegmax = g.n_src + nd;
  int iseg = 0;
  double n_lim = H3A(r.lim, 1), n_wgt = H3A(r.wgt, 1);
  int n_sd = H3A(r.sdst, 1);
  for (int js0 = 1; js0 <= ns; js0 += H3R_U) {
   int a_ns[H3R_U];
   double a_u[H3R_U], a_p1[H3R_U], a_p2[H3R_U], a_p3[H3R_U], a_p4[H3R_U], a_p5[H3R_U];
   for (int w_ = 0; w_ < H3R_U; ++w_) {
     const int jq = js0 + w_ <= ns ? js0 + w_ : ns;
     a_ns[w_] = H3A(r.nseg, jq);
     a_u[w_] = H3A(s.u, jq);
     a_p1[w_] = a_p2[w_] = a_p3[w_] = a_p4[w_] = a_p5[w_] = 0;
     if (m != H3_PCM) {
       a_p1[w_] = PC(1, jq); a_p2[w_] = PC(2, jq);
       if (m != H3_PLM) a_p3[w_] = PC(3, jq);
       if (m == H3_PQM) { a_p4[w_] = PC(4, jq); a_p5[w_] = PC(5, jq); }
     }
   }
   for (int w_ = 0; w_ < H3R_U; ++w_) {
    const int js = js0 + w_;
    if (js > ns) break;
    const int nseg = a_ns[w_];
    if (nseg == 1) {
      iseg = iseg + 1;
      const int jd = n_sd;
      const double wgt1 = n_wgt;
      { const int q_ = iseg + 1 <= nsegmax ? iseg + 1 : nsegmax; n_lim = H3A(r.lim, q_); n_wgt = H3A(r.wgt, q_); n_sd = H3A(r.sdst, q_); }
      H3_DST(jd);
      acc = acc + a_u[w_] * wgt1;
      continue;
    }
    double xil = 0.0;
    double p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
    if (m == H3_PCM) p1 = a_u[w_];
    else {
      p1 = a_p1[w_]; p2 = a_p2[w_];
      if (m != H3_PLM) p3 = a_p3[w_];
      if (m == H3_PQM) { p4 = a_p4[w_]; p5 = a_p5[w_]; }
    }
    for (int i = 1; i <= nseg; ++i) {
      iseg = iseg + 1;
      const double xir = n_lim;
      const int jd = n_sd;
      const double wgt = n_wgt;
      { const int q_ = iseg + 1 <= nsegmax ? iseg + 1 : nsegmax; n_lim = H3A(r.lim, q_); n_wgt = H3A(r.wgt, q_); n_sd = H3A(r.sdst, q_); }
      H3_DST(jd);
      if (m == H3_PCM) {
        if (xil == xir) acc = p1;
        else { acc = acc + p1 * wgt; xil = xir; }
      } else if (m == H3_PLM) {
        if (xil == xir) { acc = p2 * xir + p1; continue; }
        double v;
        if (xil == 0.0) v = 0.5 * p2 * xir + p1;
        else {
          const double b2 = 0.5 * p2;
          const double b1 = xir == 1.0 ? p1 + b2 : p1 + b2 * xir;
          v = b2 * xil + b1;
        }
        acc = acc + v * wgt;
        xil = xir;
      } else if (m == H3_PPM) {
        if (xil == xir) { acc = (p3 * xir + p2) * xir + p1; continue; }
        double v;
        if (xil == 0.0) v = ((1.0 / 3.0) * p3 * xir + 0.5 * p2) * xir + p1;
        else {
          const double b3 = (1.0 / 3.0) * p3;
          double b2, b1;
          if (xir == 1.0) { b2 = 0.5 * p2 + b3; b1 = p1 + b2; }
          else { b2 = 0.5 * p2 + b3 * xir; b1 = p1 + b2 * xir; }
          v = (b3 * xil + b2) * xil + b1;
        }
        acc = acc + v * wgt;
        xil = xir;
      } else {
        if (xil == xir) { acc = (((p5 * xir + p4) * xir + p3) * xir + p2) * xir + p1; continue; }
        double v;
        if (xil == 0.0)
          v = ((((1.0 / 5.0) * p5 * xir + 0.25 * p4) * xir + (1.0 / 3.0) * p3) * xir + 0.5 * p2) * xir + p1;
        else {
          const double b5 = (1.0 / 5.0) * p5;
          double b4, b3, b2, b1;
          if (xir == 1.0) { b4 = 0.25 * p4 + b5; b3 = (1.0 / 3.0) * p3 + b4; b2 = 0.5 * p2 + b3; b1 = p1 + b2; }
          else { b4 = 0.25 * p4 + b5 * xir; b3 = (1.0 / 3.0) * p3 + b4 * xir; b2 = 0.5 * p2 + b3 * xir; b1 = p1 + b2 * xir; }
          v = (((b5 * xil + b4) * xil + b3) * xil + b2) * xil + b1;
        }
        acc = acc + v * wgt;
        xil = xir;
      }
    }
   }
  }
#undef H3R_U
  if (cur > 0) H3A(udst, cur) = acc;
  for (int q_ = cur + 1; q_ <= nd; ++q_) H3A(udst, q_) = 0.0;
#undef H3_DST
  // near-empty destination cells at either end
  const int d1 = H3A(r.sdst, 1), dl = H3A(r.sdst, iseg);
  if (m == H3_PCM) {
    for (int jd = 1; jd <= d1 - 1; ++jd) H3A(udst, jd) = H3A(s.u, 1);
    for (int jd = dl + 1; jd <= nd; ++jd) H3A(udst, jd) = H3A(s.u, ns);
  } else {
    for (int jd = 1; jd <= d1 - 1; ++jd) H3A(udst, jd) = PC(1, 1);
    if (dl < nd) {
      double v = PC(1, ns) + PC(2, ns);
      if (m != H3_PLM) v = v + PC(3, ns);
      if (m == H3_PQM) v = v + PC(4, ns) + PC(5, ns);
      H3A(udst, dl + 1) = v;
      for (int jd = dl + 2; jd <= nd; ++jd) H3A(udst, jd) = H3A(udst, dl + 1);
    }
  }
  return H3_NOERR;
}
