// hor3map_pqm.h -- piecewise-quartic part of the batched HOR3MAP (see hor3map_core.h for the layout):
// implicit 6th-order edge / 5th-order slope estimates and the three PQM limiters.
#pragma once
#include "hor3map_core.h"

// mod_hor3map.F90:650-670
H3HD void h3_slope_ih3_coeff(double h1, double h2, double *t) {
  const double h11 = h1 * h1, h22 = h2 * h2, h12 = h1 * h2;
  const double q = 1.0 / ((h1 + h2) * (h11 + 3.0 * h12 + h22));
  t[0] = h2 * (h11 + h2 * (h1 - h2)) * q;
  t[1] = h1 * (h22 + h1 * (h2 - h1)) * q;
  t[2] = -12.0 * h12 * q;
  t[3] = -t[2];
}
H3HD void h3_edge_ih4_coeff(double h1, double h2, double *t) {   // :631-648
  const double q = 1.0 / (h1 + h2);
  t[0] = h2 * h2 * q * q;
  t[1] = h1 * h1 * q * q;
  t[2] = 2.0 * t[0] * (h2 + 2.0 * h1) * q;
  t[3] = 2.0 * t[1] * (h1 + 2.0 * h2) * q;
}

#define A6(i, j) a[((i) - 1) + 6 * ((j) - 1)]
#define B6(i, j) b[((i) - 1) + 6 * ((j) - 1)]
// mod_hor3map.F90:672-714
H3HD void h3_ih6_common(double *a, double *te, double *ts) {
  double b[36];
  for (int j = 3; j <= 6; ++j)
    for (int i = 1; i <= 5; ++i) B6(i, j) = A6(i + 1, j);
  for (int j = 1; j <= 2; ++j) {
    B6(1, j) = 1.0;
    B6(2, j) = 2.0 * A6(2, j);
    B6(3, j) = 3.0 * A6(3, j);
    B6(4, j) = 4.0 * A6(4, j);
    B6(5, j) = 5.0 * A6(5, j);
    B6(6, j) = 0.0;
  }
  for (int j = 3; j <= 6; ++j) B6(6, j) = 1.0;
  te[0] = -1.0;
  ts[0] = -1.0;
  for (int i = 1; i < 6; ++i) { te[i] = 0.0; ts[i] = 0.0; }
  h3_lu_decompose(6, a, 6);
  h3_lu_solve(6, a, 6, te);
  h3_lu_decompose(6, b, 6);
  h3_lu_solve(6, b, 6, ts);
}
// column of the moment matrix for a cell whose centre is at distance d (signed) and width h:
// rows 2..6 = -(d), -(d^2 + h^2/12), ... in the reference's factored forms (:760-776 etc.)
H3HD void h3_ih6_cell_column(double *a, int j, double a2, double h) {
  const double a2sq = a2 * a2, hsq = h * h;
  A6(1, j) = -1.0;
  A6(2, j) = a2;
  A6(3, j) = -a2sq - (1.0 / 12.0) * hsq;
  A6(4, j) = a2 * (a2sq + 0.25 * hsq);
  A6(5, j) = -a2sq * (a2sq + 0.5 * hsq) - (1.0 / 80.0) * hsq * hsq;
  A6(6, j) = a2 * (a2sq + 0.75 * hsq) * (a2sq + (1.0 / 12.0) * hsq);
}
// columns 1-2 (edge values at -hl and +hr) and the two cells adjacent to the middle edge
H3HD void h3_ih6_edge_columns(double *a, double hl, double hr, int jl_cell, int jr_cell) {
  A6(1, 1) = 1.0;
  A6(2, 1) = -hl;
  A6(3, 1) = -A6(2, 1) * hl;
  A6(4, 1) = -A6(3, 1) * hl;
  A6(5, 1) = -A6(4, 1) * hl;
  A6(6, 1) = -A6(5, 1) * hl;
  A6(1, 2) = 1.0;
  A6(2, 2) = hr;
  A6(3, 2) = A6(2, 2) * hr;
  A6(4, 2) = A6(3, 2) * hr;
  A6(5, 2) = A6(4, 2) * hr;
  A6(6, 2) = A6(5, 2) * hr;
  const int jj[2] = {jl_cell, jr_cell};
  for (int c = 0; c < 2; ++c) {
    const int j = jj[c], s = c + 1;
    A6(1, j) = -1.0;
    A6(2, j) = -0.5 * A6(2, s);
    A6(3, j) = -(1.0 / 3.0) * A6(3, s);
    A6(4, j) = -0.25 * A6(4, s);
    A6(5, j) = -(1.0 / 5.0) * A6(5, s);
    A6(6, j) = -(1.0 / 6.0) * A6(6, s);
  }
}
// mod_hor3map.F90:716-780 (asymleft), :782-845 (sym), :847-911 (asymright); h[0..3]
H3HD void h3_ih6_asymleft(const double *h, double *te, double *ts) {
  double a[36];
  h3_ih6_edge_columns(a, h[0], h[1], 3, 4);
  h3_ih6_cell_column(a, 5, -h[1] - 0.5 * h[2], h[2]);
  h3_ih6_cell_column(a, 6, -h[1] - h[2] - 0.5 * h[3], h[3]);
  h3_ih6_common(a, te, ts);
}
H3HD void h3_ih6_sym(const double *h, double *te, double *ts) {
  double a[36];
  h3_ih6_edge_columns(a, h[1], h[2], 4, 5);
  h3_ih6_cell_column(a, 3, 0.5 * h[0] + h[1], h[0]);
  h3_ih6_cell_column(a, 6, -h[2] - 0.5 * h[3], h[3]);
  h3_ih6_common(a, te, ts);
}
H3HD void h3_ih6_asymright(const double *h, double *te, double *ts) {
  double a[36];
  h3_ih6_edge_columns(a, h[2], h[3], 5, 6);
  h3_ih6_cell_column(a, 3, 0.5 * h[0] + h[1] + h[2], h[0]);
  h3_ih6_cell_column(a, 4, 0.5 * h[1] + h[2], h[1]);
  h3_ih6_common(a, te, ts);
}
#undef A6
#undef B6

// mod_hor3map.F90:1041-1306
H3HD void h3_prepare_pqm(const H3Grid &g, int col) {
  const int nc = g.nc;
  int first_index, last_index;
  int ns = h3_link_nonempty(g, col, first_index, last_index);
  if (ns < H3_NMIN_PQM) { g.n_act[col] = ns; return; }

  int jf = first_index;
  while (true) {
    int j = jf;
    double hp = H3A(g.h, j), h_max = H3A(g.h, j);
    bool end = false;
    for (int n = 1; n <= 3; ++n) {
      j = H3A(g.next, j);
      if (j == 0) { end = true; break; }
      hp = hp * H3A(g.h, j);
      h_max = h3_max(h_max, H3A(g.h, j));
    }
    if (end) break;
    if (hp > H3_HPLIM_IH6 * h3_powi(h_max, 4)) {
      jf = H3A(g.next, jf);
    } else {
      ns = ns - 1;
      if (ns < H3_NMIN_PQM) { g.n_act[col] = ns; return; }
      j = jf;
      double h_min = H3A(g.h, j);
      int j_min = j;
      for (int n = 1; n <= 3; ++n) {
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
        break;
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
        jf = jp;
        if (jf != first_index) {
          jf = H3A(g.prev, jf);
          if (jf != first_index) jf = H3A(g.prev, jf);
        }
      }
    }
  }

  int lb_ord = ns < g.left_bndr_ord ? ns : g.left_bndr_ord;
  h3_left_bndr_cond(g, col, first_index, last_index, lb_ord, ns, H3_NMIN_PQM);
  if (ns < H3_NMIN_PQM) { g.n_act[col] = ns; return; }
  int rb_ord = ns < g.right_bndr_ord ? ns : g.right_bndr_ord;
  h3_right_bndr_cond(g, col, last_index, rb_ord, ns, H3_NMIN_PQM);
  if (ns < H3_NMIN_PQM) { g.n_act[col] = ns; return; }

  h3_compact_and_weights(g, col);
  h3_continuous_edges(g, col, ns);
  for (int j = 2; j <= ns - 1; ++j)
    H3A(g.hci, j) = 2.0 / (H3A(g.h, j - 1) + 2.0 * H3A(g.h, j) + H3A(g.h, j + 1));

  double te[6], ts[6], hh[4];
  auto store = [&](int j) {
    for (int r = 1; r <= 6; ++r) {
      H3A2(g.tde, r, j, 6) = te[r - 1];
      H3A2(g.tds, r, j, 6) = ts[r - 1];
    }
  };
  auto not_dominant = [&]() {
    return h3_abs(te[0]) + h3_abs(te[1]) > 1.0 || h3_abs(ts[0]) + h3_abs(ts[1]) > 1.0;
  };
  // row 2
  {
    bool low = lb_ord < 5;
    if (!low) {
      for (int i = 0; i < 4; ++i) hh[i] = H3A(g.h, 1 + i);
      h3_ih6_asymleft(hh, te, ts);
      low = not_dominant();
    }
    if (low) {
      h3_edge_ih4_coeff(H3A(g.h, 1), H3A(g.h, 2), te);
      te[4] = 0.0; te[5] = 0.0;
      h3_slope_ih3_coeff(H3A(g.h, 1), H3A(g.h, 2), ts);
      ts[4] = 0.0; ts[5] = 0.0;
    }
    store(2);
  }
  for (int j = 3; j <= ns - 1; ++j) {
    for (int i = 0; i < 4; ++i) hh[i] = H3A(g.h, j - 2 + i);
    h3_ih6_sym(hh, te, ts);
    if (not_dominant()) {
      h3_edge_ih4_coeff(H3A(g.h, j - 1), H3A(g.h, j), te);
      te[4] = te[3]; te[3] = te[2]; te[2] = 0.0; te[5] = 0.0;
      h3_slope_ih3_coeff(H3A(g.h, j - 1), H3A(g.h, j), ts);
      ts[4] = ts[3]; ts[3] = ts[2]; ts[2] = 0.0; ts[5] = 0.0;
    }
    store(j);
  }
  {
    bool low = rb_ord < 5;
    if (!low) {
      for (int i = 0; i < 4; ++i) hh[i] = H3A(g.h, ns - 3 + i);
      h3_ih6_asymright(hh, te, ts);
      low = not_dominant();
    }
    if (low) {
      h3_edge_ih4_coeff(H3A(g.h, ns - 1), H3A(g.h, ns), te);
      te[4] = te[2]; te[5] = te[3]; te[2] = 0.0; te[3] = 0.0;
      h3_slope_ih3_coeff(H3A(g.h, ns - 1), H3A(g.h, ns), ts);
      ts[4] = ts[2]; ts[5] = ts[3]; ts[2] = 0.0; ts[3] = 0.0;
    }
    store(ns);
  }

  double hb[H3_LD], a[H3_LD * H3_LD];
  if (lb_ord > 1) {
    for (int i = 1; i <= lb_ord; ++i) hb[i - 1] = H3A(g.h, i);
    h3_edge_slope_lblu(lb_ord, hb, a);
    for (int j = 1; j <= lb_ord; ++j)
      for (int i = 1; i <= lb_ord; ++i) H3A2(g.lblu, i, j, H3_LD) = a[(i - 1) + H3_LD * (j - 1)];
  }
  if (rb_ord > 1) {
    for (int i = 1; i <= rb_ord; ++i) hb[i - 1] = H3A(g.h, ns - rb_ord + i);
    h3_edge_slope_rblu(rb_ord, hb, a);
    for (int j = 1; j <= rb_ord; ++j)
      for (int i = 1; i <= rb_ord; ++i) H3A2(g.rblu, i, j, H3_LD) = a[(i - 1) + H3_LD * (j - 1)];
  }
  g.n_act[col] = ns;
  g.lb_act[col] = lb_ord;
  g.rb_act[col] = rb_ord;
}

// mod_hor3map.F90:1765-1870
H3HD void h3_pqm_edge_slope_values(const H3Grid &g, const H3Src &s, int col) {
  const int nc = g.nc;
  const int ns = g.n_act[col], lb_ord = g.lb_act[col], rb_ord = g.rb_act[col];
  const size_t plane = (size_t)(g.n_src + 1) * nc;
  double *uedge = s.wk, *uslope = s.wk + plane, *gam = s.wk + 2 * plane;
  double x[H3_LD], lu[H3_LD * H3_LD];
  if (lb_ord == 1) {
    H3A(uedge, 1) = H3A(s.u, 1);
    H3A(uslope, 1) = 0.0;
  } else {
    for (int i = 1; i <= lb_ord; ++i) x[i - 1] = H3A(s.u, i);
    for (int j = 1; j <= lb_ord; ++j)
      for (int i = 1; i <= lb_ord; ++i) lu[(i - 1) + H3_LD * (j - 1)] = H3A2(g.lblu, i, j, H3_LD);
    h3_lu_solve(lb_ord, lu, H3_LD, x);
    H3A(uedge, 1) = x[0];
    H3A(uslope, 1) = x[1];
  }
  if (rb_ord == 1) {
    H3A(uedge, ns + 1) = H3A(s.u, ns);
    H3A(uslope, ns + 1) = 0.0;
  } else {
    for (int i = 1; i <= rb_ord; ++i) x[i - 1] = H3A(s.u, ns - rb_ord + i);
    for (int j = 1; j <= rb_ord; ++j)
      for (int i = 1; i <= rb_ord; ++i) lu[(i - 1) + H3_LD * (j - 1)] = H3A2(g.rblu, i, j, H3_LD);
    h3_lu_solve(rb_ord, lu, H3_LD, x);
    H3A(uedge, ns + 1) = x[0];
    H3A(uslope, ns + 1) = x[1];
  }
  for (int pass = 0; pass < 2; ++pass) {
    const double *t = pass == 0 ? g.tde : g.tds;
    double *ue = pass == 0 ? uedge : uslope;
    H3A(gam, 1) = 0.0;
    for (int j = 2; j <= ns; ++j) {
      const int j0 = j == 2 ? 1 : (j == ns ? ns - 3 : j - 2);   // first cell of the four-cell stencil
      const double rhs = H3A2(t, 3, j, 6) * H3A(s.u, j0) + H3A2(t, 4, j, 6) * H3A(s.u, j0 + 1) +
                         H3A2(t, 5, j, 6) * H3A(s.u, j0 + 2) + H3A2(t, 6, j, 6) * H3A(s.u, j0 + 3);
      const double t1 = H3A2(t, 1, j, 6);
      const double bei = 1.0 / (1.0 - t1 * H3A(gam, j - 1));
      H3A(ue, j) = (rhs - t1 * H3A(ue, j - 1)) * bei;
      H3A(gam, j) = H3A2(t, 2, j, 6) * bei;
    }
    for (int j = ns; j >= 2; --j) H3A(ue, j) = H3A(ue, j) - H3A(gam, j) * H3A(ue, j + 1);
  }
  for (int j = 1; j <= ns; ++j) {
    H3A(s.uel, j) = H3A(uedge, j);
    H3A(s.uer, j) = H3A(uedge, j + 1);
    H3A(s.usl, j) = H3A(uslope, j) * H3A(g.h, j);
    H3A(s.usr, j) = H3A(uslope, j + 1) * H3A(g.h, j);
  }
}

// the finite-difference limiting of an unsmooth interior cell (:2135-2159, :2408-2434); returns sl, sr, sc
H3HD void h3_pqm_limit_cell(const H3Grid &g, const H3Src &s, int col, int j, double &sl, double &sr, double &sc) {
  const int nc = g.nc;
  const double um = H3A(s.u, j - 1), u0 = H3A(s.u, j), up = H3A(s.u, j + 1);
  sl = 2.0 * (u0 - um) * H3A(g.hi, j);
  sr = 2.0 * (up - u0) * H3A(g.hi, j);
  sc = (up - um) * H3A(g.hci, j);
  sc = h3_sign(h3_min(h3_min(h3_abs(sl), h3_abs(sr)), h3_abs(sc)), sc);
  if (sl * sr > 0.0) {
    const double el = H3A(s.uel, j), er = H3A(s.uer, j);
    if ((um - el) * (u0 - el) > 0.0)
      H3A(s.uel, j) = u0 - h3_sign(h3_min(0.5 * H3A(g.h, j) * h3_abs(sc), h3_abs(el - u0)), sc);
    if ((up - er) * (u0 - er) > 0.0)
      H3A(s.uer, j) = u0 + h3_sign(h3_min(0.5 * H3A(g.h, j) * h3_abs(sc), h3_abs(er - u0)), sc);
    if (H3A(s.usl, j) * sc < 0.0) H3A(s.usl, j) = 0.0;
    if (H3A(s.usr, j) * sc < 0.0) H3A(s.usr, j) = 0.0;
  } else {
    H3A(s.uel, j) = u0;
    H3A(s.uer, j) = u0;
    H3A(s.usl, j) = 0.0;
    H3A(s.usr, j) = 0.0;
  }
}

// removal of inflection points that contradict the limited slope (:2170-2265, :2453-2552)
H3HD void h3_pqm_fix_inflexion(const H3Src &s, int nc, int col, int j, double sl, double sr, double sc,
                               double u_eps, double uu_eps) {
  const double u0 = H3A(s.u, j);
  double el = H3A(s.uel, j), er = H3A(s.uer, j), usl = H3A(s.usl, j), usr = H3A(s.usr, j);
  const double a0 = usl;
  const double a1 = 2.0 * (30.0 * u0 - 18.0 * el - 12.0 * er - 4.5 * usl + 1.5 * usr);
  const double a2 = 3.0 * (-60.0 * u0 + 32.0 * el + 28.0 * er + 6.0 * usl - 4.0 * usr);
  const double a3 = 4.0 * (30.0 * u0 - 15.0 * (el + er) - 2.5 * (usl - usr));
  const double b0 = a1, b1 = 2.0 * a2, b2 = 3.0 * a3;
  bool incon = false;
  const double q1 = b0 * b2;
  const double q2 = b1 * b1 - 4.0 * q1;
#define DERIV(xi) ((a0 + (xi) * (a1 + (xi) * (a2 + (xi) * a3))) * sc < 0.0)
  if (q2 > 0.0) {
    if (b0 * (b0 + b1 + b2) < 0.0) {
      if (h3_abs(b2) < u_eps) {
        if (h3_abs(b1) > u_eps) {
          const double xi = -b0 / b1;
          if (DERIV(xi)) incon = true;
        }
      } else {
        const double q3 = 0.5 / b2;
        const double sq = __builtin_sqrt(q2);
        double xi = -(b1 + sq) * q3;
        if (xi > 0.0 && xi < 1.0) {
          if (DERIV(xi)) incon = true;
        } else {
          xi = -(b1 - sq) * q3;
          if (DERIV(xi)) incon = true;
        }
      }
    } else if (q1 > uu_eps) {
      const double q3 = 0.5 / b2;
      const double sq = __builtin_sqrt(q2);
      double xi = -(b1 + sq) * q3;
      if (DERIV(xi)) incon = true;
      else {
        xi = -(b1 - sq) * q3;
        if (DERIV(xi)) incon = true;
      }
    }
  }
#undef DERIV
  if (!incon) return;
  if (h3_abs(sl) < h3_abs(sr)) {
    usl = (10.0 / 3.0) * u0 - (8.0 / 3.0) * el - (2.0 / 3.0) * er;
    if (usl * sc < 0.0) {
      usl = 0.0;
      er = 5.0 * u0 - 4.0 * el;
      usr = 20.0 * (u0 - el);
    } else {
      usr = 4.0 * el + 6.0 * er - 10.0 * u0;
      if (usr * sc < 0.0) {
        usr = 0.0;
        el = 2.5 * u0 - 1.5 * er;
        usl = (10.0 / 3.0) * (er - u0);
      }
    }
  } else {
    usr = (8.0 / 3.0) * er + (2.0 / 3.0) * el - (10.0 / 3.0) * u0;
    if (usr * sc < 0.0) {
      usr = 0.0;
      el = 5.0 * u0 - 4.0 * er;
      usl = 20.0 * (er - u0);
    } else {
      usl = 10.0 * u0 - 4.0 * er - 6.0 * el;
      if (usl * sc < 0.0) {
        usl = 0.0;
        er = 2.5 * u0 - 1.5 * el;
        usr = (10.0 / 3.0) * (u0 - el);
      }
    }
  }
  H3A(s.uel, j) = el; H3A(s.uer, j) = er; H3A(s.usl, j) = usl; H3A(s.usr, j) = usr;
}

// boundary cells (:2267-2335; the posdef variant :2858-2932 differs in two branches)
H3HD void h3_pqm_boundary(const H3Grid &g, const H3Src &s, int col, bool posdef) {
  const int nc = g.nc, ns = g.n_act[col];
  const double u1 = H3A(s.u, 1), un = H3A(s.u, ns);
  if (s.pc_left || (H3A(s.u, 2) - H3A(s.uer, 1)) * (u1 - H3A(s.uer, 1)) > 0.0) {
    H3A(s.uel, 1) = u1; H3A(s.uer, 1) = u1; H3A(s.usl, 1) = 0.0; H3A(s.usr, 1) = 0.0;
  } else {
    const double sl = 2.0 * (H3A(s.u, 3) - H3A(s.u, 2)) / (H3A(g.h, 2) + H3A(g.h, 3));
    const double b = u1 + (1.0 / 3.0) * sl * H3A(g.h, 1);
    if (sl > 0) {
      H3A(s.uer, 1) = h3_max(u1, h3_min(H3A(s.uel, 2), b));
      if (posdef) {
        H3A(s.uel, 1) = h3_max(h3_min(u1, 0.0), 0.5 * (3.0 * u1 - H3A(s.uer, 1)));
        H3A(s.uer, 1) = 3.0 * u1 - 2.0 * H3A(s.uel, 1);
      } else {
        H3A(s.uel, 1) = 0.5 * (3.0 * u1 - H3A(s.uer, 1));
      }
    } else {
      H3A(s.uer, 1) = h3_min(u1, h3_max(H3A(s.uel, 2), b));
      H3A(s.uel, 1) = 0.5 * (3.0 * u1 - H3A(s.uer, 1));
    }
    H3A(s.usl, 1) = 6.0 * u1 - 4.0 * H3A(s.uel, 1) - 2.0 * H3A(s.uer, 1);
    H3A(s.usr, 1) = 2.0 * H3A(s.uel, 1) + 4.0 * H3A(s.uer, 1) - 6.0 * u1;
  }
  if (s.pc_right || (un - H3A(s.uel, ns)) * (H3A(s.u, ns - 1) - H3A(s.uel, ns)) > 0.0) {
    H3A(s.uel, ns) = un; H3A(s.uer, ns) = un; H3A(s.usl, ns) = 0.0; H3A(s.usr, ns) = 0.0;
  } else {
    const double sl = 2.0 * (H3A(s.u, ns - 1) - H3A(s.u, ns - 2)) / (H3A(g.h, ns - 2) + H3A(g.h, ns - 1));
    const double b = un - (1.0 / 3.0) * sl * H3A(g.h, ns);
    if (sl > 0) {
      H3A(s.uel, ns) = h3_min(un, h3_max(H3A(s.uer, ns - 1), b));
      H3A(s.uer, ns) = 0.5 * (3.0 * un - H3A(s.uel, ns));
    } else {
      H3A(s.uel, ns) = h3_max(un, h3_min(H3A(s.uer, ns - 1), b));
      if (posdef) {
        H3A(s.uer, ns) = h3_max(h3_min(un, 0.0), 0.5 * (3.0 * un - H3A(s.uel, ns)));
        H3A(s.uel, ns) = 3.0 * un - 2.0 * H3A(s.uer, ns);
      } else {
        H3A(s.uer, ns) = 0.5 * (3.0 * un - H3A(s.uel, ns));
      }
    }
    H3A(s.usl, ns) = 6.0 * un - 4.0 * H3A(s.uel, ns) - 2.0 * H3A(s.uer, ns);
    H3A(s.usr, ns) = 2.0 * H3A(s.uel, ns) + 4.0 * H3A(s.uer, ns) - 6.0 * un;
  }
}

// limit_pqm_monotonic (mod_hor3map.F90:2119-2337)
H3HD void h3_limit_pqm_monotonic(const H3Grid &g, const H3Src &s, int col) {
  const int nc = g.nc, ns = g.n_act[col];
  const size_t plane = (size_t)(g.n_src + 1) * nc;
  double *SL = s.wk, *SR = s.wk + plane, *SC = s.wk + 2 * plane;
  for (int j = 2; j <= ns - 1; ++j) {
    double sl, sr, sc;
    h3_pqm_limit_cell(g, s, col, j, sl, sr, sc);
    H3A(SL, j) = sl; H3A(SR, j) = sr; H3A(SC, j) = sc;
  }
  h3_ppm_edge_consistency(s, nc, col, ns);
  const double u_eps = s.u_eps[col], uu_eps = s.uu_eps[col];
  for (int j = 2; j <= ns - 1; ++j)
    h3_pqm_fix_inflexion(s, nc, col, j, H3A(SL, j), H3A(SR, j), H3A(SC, j), u_eps, uu_eps);
  h3_pqm_boundary(g, s, col, false);
}

// limit_pqm_non_oscillatory (:2339-2624) and ..._posdef (:2626-2934)
H3HD void h3_limit_pqm_non_oscillatory(const H3Grid &g, const H3Src &s, int col, bool posdef) {
  const int nc = g.nc, ns = g.n_act[col];
  const size_t plane = (size_t)(g.n_src + 1) * nc;
  double *SL = s.wk, *SR = s.wk + plane, *SC = s.wk + 2 * plane, *D2 = s.wk + 3 * plane;
  const double u_eps = s.u_eps[col], uu_eps = s.uu_eps[col];
  for (int j = 1; j <= ns; ++j) H3A(D2, j) = H3A(s.uel, j) - 2.0 * H3A(s.u, j) + H3A(s.uer, j);
  // smooth(j) is re-derived from d2 where needed (d2 itself is not modified below)
#define SMOOTH(j) (H3A(D2, (j) - 1) * H3A(D2, j) >= 0.0 && H3A(D2, j) * H3A(D2, (j) + 1) >= 0.0)
  for (int j = 2; j <= ns - 1; ++j) {
    if (SMOOTH(j)) {
      const double u0 = H3A(s.u, j);
      if (posdef) {
        const double min_u_0 = h3_min(u0, 0.0);
        H3A(s.uel, j) = h3_max(H3A(s.uel, j), min_u_0);
        H3A(s.uer, j) = h3_max(H3A(s.uer, j), min_u_0);
      }
      const double el = H3A(s.uel, j), er = H3A(s.uer, j);
      const double sl = 6.0 * u0 - 4.0 * el - 2.0 * er;
      const double sr = 2.0 * el + 4.0 * er - 6.0 * u0;
      H3A(SL, j) = sl; H3A(SR, j) = sr;
      if (sl < 0.0 && sr > 0.0) {
        if (posdef) {
          const double min_u_0 = h3_min(u0, 0.0);
          const double a2 = 0.5 * (sr - sl);
          if (a2 * el - 0.25 * sl * sl < a2 * min_u_0) {
            const double q1 = 3.0 * u0 / (3.0 * sl * sr + 4.0 * a2 * a2);
            const double nel = sl * sl * q1, ner = sr * sr * q1;
            H3A(s.uel, j) = nel;
            H3A(s.uer, j) = ner;
            H3A(s.usl, j) = 6.0 * u0 - 4.0 * nel - 2.0 * ner;
            H3A(s.usr, j) = 2.0 * nel + 4.0 * ner - 6.0 * u0;
          } else {
            H3A(s.usl, j) = sl;
            H3A(s.usr, j) = sr;
          }
        } else {
          H3A(s.usl, j) = sl;
          H3A(s.usr, j) = sr;
        }
      } else {
        const double usl = H3A(s.usl, j), usr = H3A(s.usr, j);
        const double b0 = 2.0 * (30.0 * u0 - 18.0 * el - 12.0 * er - 4.5 * usl + 1.5 * usr);
        const double b1 = 6.0 * (-60.0 * u0 + 32.0 * el + 28.0 * er + 6.0 * usl - 4.0 * usr);
        const double b2 = 12.0 * (30.0 * u0 - 15.0 * (el + er) - 2.5 * (usl - usr));
        const double q1 = b0 * b2;
        const double q2 = b1 * b1 - 4.0 * q1;
        if (q2 > 0.0 && (b0 * (b0 + b1 + b2) < 0.0 || q1 > uu_eps)) {
          H3A(s.usl, j) = sl;
          H3A(s.usr, j) = sr;
        }
      }
    } else {
      double sl, sr, sc;
      h3_pqm_limit_cell(g, s, col, j, sl, sr, sc);
      H3A(SL, j) = sl; H3A(SR, j) = sr; H3A(SC, j) = sc;
    }
  }
  for (int j = 3; j <= ns - 1; ++j)
    if ((H3A(s.uel, j) - H3A(s.uer, j - 1)) * (H3A(s.u, j) - H3A(s.u, j - 1)) < 0.0) {
      if (SMOOTH(j - 1)) H3A(s.uel, j) = H3A(s.uer, j - 1);
      else if (SMOOTH(j)) H3A(s.uer, j - 1) = H3A(s.uel, j);
      else {
        H3A(s.uel, j) = 0.5 * (H3A(s.uer, j - 1) + H3A(s.uel, j));
        H3A(s.uer, j - 1) = H3A(s.uel, j);
      }
    }
  for (int j = 2; j <= ns - 1; ++j)
    if (!SMOOTH(j)) h3_pqm_fix_inflexion(s, nc, col, j, H3A(SL, j), H3A(SR, j), H3A(SC, j), u_eps, uu_eps);
#undef SMOOTH
  h3_pqm_boundary(g, s, col, posdef);
}

// polycoeff_pqm (mod_hor3map.F90:2936-2960)
H3HD void h3_polycoeff_pqm(const H3Grid &g, const H3Src &s, int col) {
  const int nc = g.nc, np = g.p_ord + 1, ns = g.n_act[col];
  for (int j = 1; j <= ns; ++j) {
    const double u0 = H3A(s.u, j), el = H3A(s.uel, j), er = H3A(s.uer, j), usl = H3A(s.usl, j),
                 usr = H3A(s.usr, j);
    PC(1, j) = el;
    PC(2, j) = usl;
    PC(3, j) = 30.0 * u0 - 18.0 * el - 12.0 * er - 4.5 * usl + 1.5 * usr;
    PC(4, j) = -60.0 * u0 + 32.0 * el + 28.0 * er + 6.0 * usl - 4.0 * usr;
    PC(5, j) = 30.0 * u0 - 15.0 * (el + er) - 2.5 * (usl - usr);
  }
}

H3HD void h3_reconstruct_pqm(const H3Grid &g, const H3Src &s, int col) {
  h3_pqm_edge_slope_values(g, s, col);
  switch (s.limiting) {
    case H3_MONOTONIC: h3_limit_pqm_monotonic(g, s, col); break;
    case H3_NON_OSCILLATORY: h3_limit_pqm_non_oscillatory(g, s, col, false); break;
    case H3_NON_OSCILLATORY_POSDEF: h3_limit_pqm_non_oscillatory(g, s, col, true); break;
    default: break;
  }
  h3_polycoeff_pqm(g, s, col);
}
