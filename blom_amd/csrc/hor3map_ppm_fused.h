// hor3map_ppm_fused.h -- PPM reconstruction in two sweeps per column instead of the reference's nine
// array passes (edge solve forward/backward, copy to cell edges, three limiter passes, boundary cells,
// positive-definite pass, coefficients: ~43 planes of traffic per call, which is what bounds the kernel).
//
// After the forward elimination everything else is a stencil of reach one in the layer index:
//   * back substitution of edge j needs edge j+1,
//   * the cell limiter (:1886-1904) needs the raw edges of cell j and u(j-1..j+1); its non-oscillatory
//     switch (:1953) the raw second derivatives of cells j-1, j, j+1,
//   * the edge-consistency step (:1908-1914) touches the disjoint pairs (uer(j-1), uel(j)), j = 3..ns-1,
//   * overshoot removal (:1917-1924), positive definiteness (:2082-2096) and the coefficients (:2110-2115)
//     are per cell.
// So they ride on the backward sweep with a window of three cells in registers, each cell being stored
// once when its two edges are final.  The arithmetic of every step is the reference's, statement by
// statement; only the order in which independent cells are visited differs, and h(j)'s inverse and the
// centred-slope factor are recomputed from the widths with the expressions that produced the stored ones
// (c1/h, c2/(h(j-1)+c2*h(j)+h(j+1)), :1439, :1477) -- the same bits.  ~18 planes of traffic.
#pragma once
#include "hor3map_core.h"

#define H3F_U 8

struct H3Cell {
  double u, el, er;     // mean, left and right edge value
  bool lim;             // interior cell on which the limiter acts (always, or by the non-oscillatory switch)
};

// limit_ppm_interior_*: the per-cell part (:1886-1904)
H3HD void h3f_limit_cell(H3Cell &c, double um, double up, double h, double hi, double hci) {
  const double u0 = c.u;
  const double sl = 2.0 * (u0 - um) * hi;
  const double sr = 2.0 * (up - u0) * hi;
  if (sl * sr > 0.0) {
    double sc = (up - um) * hci;
    sc = h3_sign(h3_min(h3_min(h3_abs(sl), h3_abs(sr)), h3_abs(sc)), sc);
    const double el = c.el, er = c.er;
    if ((um - el) * (u0 - el) > 0.0) c.el = u0 - h3_sign(h3_min(0.5 * h * h3_abs(sc), h3_abs(el - u0)), sc);
    if ((up - er) * (u0 - er) > 0.0) c.er = u0 + h3_sign(h3_min(0.5 * h * h3_abs(sc), h3_abs(er - u0)), sc);
  } else {
    c.el = u0;
    c.er = u0;
  }
}
H3HD void h3f_no_overshoot(H3Cell &c) {                         // :1917-1924
  const double d = c.er - c.el;
  const double q = d * (2.0 * c.u - c.el - c.er);
  const double r = (1.0 / 3.0) * d * d;
  if (q > r) c.el = 3.0 * c.u - 2.0 * c.er;
  else if (-r > q) c.er = 3.0 * c.u - 2.0 * c.el;
}
H3HD void h3f_posdef(H3Cell &c) {                               // :2083-2095
  const double u0 = c.u;
  const double min_u_0 = h3_min(u0, 0.0);
  c.el = h3_max(c.el, min_u_0);
  c.er = h3_max(c.er, min_u_0);
  const double sl = 2.0 * (3.0 * u0 - 2.0 * c.el - c.er);
  const double a2 = 3.0 * (c.el - 2.0 * u0 + c.er);
  const double sr = sl + 2.0 * a2;
  if (sl < 0.0 && sr > 0.0)
    if (a2 * c.el - 0.25 * sl * sl < a2 * min_u_0) {
      const double q = 3.0 * u0 / (3.0 * sl * sr + 4.0 * a2 * a2);
      c.el = sl * sl * q;
      c.er = sr * sr * q;
    }
}

H3HD void h3_reconstruct_ppm_fused(const H3Grid &g, const H3Src &s, int col) {
  const int nc = g.nc, np = g.p_ord + 1;
  const int ns = g.n_act[col], lb_ord = g.lb_act[col], rb_ord = g.rb_act[col];
  const int lim = s.limiting;
  const bool limited = lim != H3_NO_LIMITING;
  const bool nonosc = lim == H3_NON_OSCILLATORY || lim == H3_NON_OSCILLATORY_POSDEF;
  const bool posdef = lim == H3_NON_OSCILLATORY_POSDEF;
  double *uedge = s.wk, *gam = s.wk + (size_t)(g.n_src + 1) * nc;

  // ---- boundary edge values (:1724-1740; the right one tests lb_ord, as the reference does) -------------
  double e_first, e_last;
  if (lb_ord == 1) {
    e_first = H3A(s.u, 1);
    e_last = H3A(s.u, ns);
  } else {
    e_first = h3_bndr_first(g.lblu, s.u, nc, col, lb_ord, 0);          // order a compile-time constant inside: no scratch
    e_last = h3_bndr_first(g.rblu, s.u, nc, col, rb_ord, ns - rb_ord);
  }

  // ---- sweep 1, ascending: forward elimination of the edge system (:1743-1754) -------------------------
  {
    double ue_prev = e_first, gam_prev = 0.0;
    double h1 = H3A(g.h, 1), u1 = H3A(s.u, 1);
    H3A(uedge, 1) = e_first;
    // H3F_U levels' loads are issued before the first is used (one dependent load per level is what bounds a
    // thread-per-column walk); the arithmetic and its order are unchanged
    for (int j0 = 2; j0 <= ns; j0 += H3F_U) {
     double ah[H3F_U], au[H3F_U];
     for (int w_ = 0; w_ < H3F_U; ++w_) {
       const int jq = j0 + w_ <= ns ? j0 + w_ : ns;
       ah[w_] = H3A(g.h, jq);
       au[w_] = H3A(s.u, jq);
     }
     for (int w_ = 0; w_ < H3F_U; ++w_) {
      const int j = j0 + w_;
      if (j > ns) break;
      const double h2 = ah[w_], u2 = au[w_];
      // edge_ih4_coeff(h(j-1:j)) (:642-646)
      const double q = 1.0 / (h1 + h2);
      const double t1 = h2 * h2 * q * q, t2 = h1 * h1 * q * q;
      const double t3 = 2.0 * t1 * (h2 + 2.0 * h1) * q, t4 = 2.0 * t2 * (h1 + 2.0 * h2) * q;
      const double rhs = t3 * u1 + t4 * u2;
      const double bei = 1.0 / (1.0 - t1 * gam_prev);
      ue_prev = (rhs - t1 * ue_prev) * bei;
      gam_prev = t2 * bei;
      H3A(uedge, j) = ue_prev;
      H3A(gam, j) = gam_prev;
      h1 = h2;
      u1 = u2;
     }
    }
  }

  // ---- sweep 2, descending: back substitution + everything that follows --------------------------------
  // e(j) = final edge value j.  Window: cells j+1 ("hi", waiting for its left edge to become final),
  // j ("mid", being limited); the non-oscillatory switch looks one edge further down.
  auto store = [&](int j, H3Cell &c) {
    if (posdef) h3f_posdef(c);
    H3A(s.uel, j) = c.el;
    H3A(s.uer, j) = c.er;
    PC(1, j) = c.el;                                               // :2111-2114
    PC(2, j) = 6.0 * c.u - 4.0 * c.el - 2.0 * c.er;
    PC(3, j) = 3.0 * (c.el - 2.0 * c.u + c.er);
  };
  auto edge = [&](int j, double e_above) {                          // final edge j given final edge j+1
    return j >= 2 ? H3A(uedge, j) - H3A(gam, j) * e_above : e_first;
  };

  double e_hi = e_last;                          // e(j+1)
  double e_mid = edge(ns, e_last);               // e(j)      (ns >= 3 here)
  double e_lo = edge(ns - 1, e_mid);             // e(j-1)
  double u_hi = 0.0, u_mid = H3A(s.u, ns), u_lo = H3A(s.u, ns - 1);        // u(j+1), u(j), u(j-1)
  double h_hi = 0.0, h_mid = H3A(g.h, ns), h_lo = H3A(g.h, ns - 1);        // h(j+1), h(j), h(j-1)
  double d2_hi = 0.0;                            // raw second derivative of cell j+1
  H3Cell hi{0.0, 0.0, 0.0, false};
  for (int j0 = ns; j0 >= 1; j0 -= H3F_U) {
   // what the window takes in at level j: u, h and the edge system's row of level j-2
   double bu[H3F_U], bh[H3F_U], be[H3F_U], bg[H3F_U];
   for (int q = 0; q < H3F_U; ++q) {
     const int jq = j0 - q - 2 >= 1 ? j0 - q - 2 : 1;
     bu[q] = H3A(s.u, jq);
     bh[q] = H3A(g.h, jq);
     be[q] = H3A(uedge, jq);
     bg[q] = H3A(gam, jq >= 2 ? jq : 2);
   }
   for (int q = 0; q < H3F_U; ++q) {
    const int j = j0 - q;
    if (j < 1) break;
    // raw parabola of cell j and of the cell below it
    H3Cell mid{u_mid, e_mid, e_hi, false};
    const double d2_mid = e_mid - 2.0 * u_mid + e_hi;
    const double d2_lo = j >= 2 ? e_lo - 2.0 * u_lo + e_mid : 0.0;
    if (limited) {
      if (j == ns) {                                                // right boundary cell (:2041-2068)
        if (s.pc_right) { mid.el = u_mid; mid.er = u_mid; }
        else if ((u_mid - mid.el) * (u_lo - mid.el) > 0.0) { mid.el = u_mid; mid.er = u_mid; }
        else {
          const double sl = 2.0 * (u_lo - H3A(s.u, ns - 2)) / (H3A(g.h, ns - 2) + h_lo);
          const double b = u_mid - (1.0 / 3.0) * sl * h_mid;
          if (sl > 0) mid.el = h3_min(u_mid, h3_max(mid.el, b));
          else mid.el = h3_max(u_mid, h3_min(mid.el, b));
          mid.er = 0.5 * (3.0 * u_mid - mid.el);
        }
      } else if (j == 1) {                                          // left boundary cell (:2012-2039)
        if (s.pc_left) { mid.el = u_mid; mid.er = u_mid; }
        else if ((u_hi - mid.er) * (u_mid - mid.er) > 0.0) { mid.el = u_mid; mid.er = u_mid; }
        else {
          const double sl = 2.0 * (H3A(s.u, 3) - u_hi) / (h_hi + H3A(g.h, 3));
          const double b = u_mid + (1.0 / 3.0) * sl * h_mid;
          if (sl > 0) mid.er = h3_max(u_mid, h3_min(mid.er, b));
          else mid.er = h3_min(u_mid, h3_max(mid.er, b));
          mid.el = 0.5 * (3.0 * u_mid - mid.er);
        }
      } else {                                                      // interior cell
        mid.lim = !nonosc || d2_lo * d2_mid < 0.0 || d2_mid * d2_hi < 0.0;      // :1953
        if (mid.lim)
          h3f_limit_cell(mid, u_lo, u_hi, h_mid, 1.0 / h_mid, 2.0 / (h_lo + 2.0 * h_mid + h_hi));
      }
      // edge consistency between cells j and j+1 (the reference's loop index j+1 in 3..ns-1)
      if (j >= 2 && j <= ns - 2)
        if ((hi.el - mid.er) * (u_hi - u_mid) < 0.0) {
          hi.el = 0.5 * (mid.er + hi.el);
          mid.er = hi.el;
        }
      // cell j+1 is complete now
      if (j + 1 <= ns - 1 && hi.lim) h3f_no_overshoot(hi);
    }
    if (j + 1 <= ns) store(j + 1, hi);
    // shift the window down
    hi = mid;
    d2_hi = d2_mid;
    u_hi = u_mid; u_mid = u_lo;
    h_hi = h_mid; h_mid = h_lo;
    e_hi = e_mid; e_mid = e_lo;
    if (j >= 3) {
      u_lo = bu[q];
      h_lo = bh[q];
      e_lo = j - 2 >= 2 ? be[q] - bg[q] * e_mid : e_first;
    }
   }
  }
  store(1, hi);
}
