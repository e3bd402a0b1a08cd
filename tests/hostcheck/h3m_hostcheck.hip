// TEST INFRASTRUCTURE.  Runs the per-column routines of blom_amd/csrc/hor3map_core.h on the HOST
// (they are __host__ __device__) with the same [level][column] layout the kernels use, so that the
// logic can be checked against the reference in the CPU test suite where no GPU exists.  It is
// never linked into libblomgpu.so and nothing in the product calls it.
#include "../../blom_amd/csrc/hor3map_core.h"
#include "../../blom_amd/csrc/hor3map_pqm.h"
#include "../../blom_amd/csrc/hor3map_ppm_fused.h"
#include <vector>
#include <cstring>

namespace {
template <class T> T *alloc(std::vector<std::vector<char>> &keep, size_t n) {
  keep.emplace_back(n * sizeof(T) ? n * sizeof(T) : 8, 0);   // exact size: an overrun is visible to a sanitizer build
  return (T *)keep.back().data();
}
void to_dev(const double *a, double *t, int m, int nc) {
  for (int c = 0; c < nc; ++c)
    for (int r = 0; r < m; ++r) t[(size_t)r * nc + c] = a[(size_t)c * m + r];
}
void from_dev(const double *t, double *a, int m, int nc) {
  for (int c = 0; c < nc; ++c)
    for (int r = 0; r < m; ++r) a[(size_t)c * m + r] = t[(size_t)r * nc + c];
}
}  // namespace

extern "C" void h3m_hostcheck_run(int method, int lb_ord, int rb_ord, int limiting, int pc_l, int pc_r, int ncol,
                                  int n_src, int n_dst, int n_grd, int regrid_method, const double *x_src,
                                  const double *u_src, const double *x_dst, const double *u_grd, double missing,
                                  double *polycoeff, double *u_dst, double *x_grd, int *errs, int *n_act,
                                  int *m_act) {
  std::vector<std::vector<char>> keep;
  const size_t nc = ncol, n = n_src, nd = n_dst, ng = n_grd;
  H3Grid g{};
  g.nc = ncol; g.n_src = n_src; g.method = method;
  int p_ord = method == H3_PCM ? 0 : method == H3_PLM ? 1 : method == H3_PPM ? 2 : 4;
  const int emax = method == H3_PQM ? H3_EB_MAX_PQM : H3_EB_MAX_PPM;
  auto clampo = [&](int o) { return o == 0 ? emax : (o < 1 ? 1 : (o > emax ? emax : o)); };
  g.left_bndr_ord = clampo(lb_ord); g.right_bndr_ord = clampo(rb_ord);
  g.p_ord = p_ord; g.ncoef = p_ord + 2;
  g.xin = alloc<double>(keep, (n + 1) * nc); g.x_eps = alloc<double>(keep, nc);
  g.x_edge = alloc<double>(keep, (n + 1) * nc);
  g.h = alloc<double>(keep, n * nc); g.hi = alloc<double>(keep, n * nc); g.hci = alloc<double>(keep, n * nc);
  g.w = alloc<double>(keep, n * nc);
  g.tde = alloc<double>(keep, 6 * n * nc); g.tds = alloc<double>(keep, 6 * n * nc);
  g.lblu = alloc<double>(keep, 36 * nc); g.rblu = alloc<double>(keep, 36 * nc);
  g.sdi = alloc<int>(keep, n * nc);
  g.n_act = alloc<int>(keep, nc); g.m_act = alloc<int>(keep, nc); g.lb_act = alloc<int>(keep, nc);
  g.rb_act = alloc<int>(keep, nc); g.prepared = alloc<int>(keep, nc); g.err = alloc<int>(keep, nc);
  g.prev = alloc<int>(keep, n * nc); g.next = alloc<int>(keep, n * nc);
  H3Src s{};
  s.limiting = limiting; s.pc_left = pc_l != 0; s.pc_right = pc_r != 0;
  s.u = alloc<double>(keep, n * nc); s.uel = alloc<double>(keep, n * nc); s.uer = alloc<double>(keep, n * nc);
  s.usl = alloc<double>(keep, n * nc); s.usr = alloc<double>(keep, n * nc);
  s.pc = alloc<double>(keep, (p_ord + 1) * n * nc);
  s.u_range = alloc<double>(keep, nc); s.u_eps = alloc<double>(keep, nc); s.uu_eps = alloc<double>(keep, nc);
  s.reconstructed = alloc<int>(keep, nc); s.err = g.err;
  s.wk = alloc<double>(keep, 4 * (n + 1) * nc);
  H3Map r{};
  r.n_dst = n_dst;
  r.lim = alloc<double>(keep, (n + nd) * nc); r.wgt = alloc<double>(keep, (n + nd) * nc);
  r.nseg = alloc<int>(keep, n * nc); r.sdst = alloc<int>(keep, (n + nd) * nc);
  r.prepared = alloc<int>(keep, nc); r.err = g.err; r.hdst = alloc<double>(keep, nd * nc);

  double *uin = alloc<double>(keep, n * nc), *xd = alloc<double>(keep, (nd + 1) * nc);
  double *ug = alloc<double>(keep, ng * nc), *xg = alloc<double>(keep, ng * nc);
  double *pco = alloc<double>(keep, (p_ord + 1) * n * nc), *ud = alloc<double>(keep, nd * nc);
  to_dev(x_src, g.xin, n_src + 1, ncol);
  to_dev(u_src, uin, n_src, ncol);
  to_dev(x_dst, xd, n_dst + 1, ncol);
  to_dev(u_grd, ug, n_grd, ncol);
  to_dev(u_dst, ud, n_dst, ncol);
  to_dev(x_grd, xg, n_grd, ncol);
  if (regrid_method == 0) regrid_method = H3_REGRID_METHOD_1;
  for (int col = 0; col < ncol; ++col) {
    int *e = errs + 6 * col;
    e[0] = h3_prepare_reconstruction(g, col);
    e[1] = h3_reconstruct(g, s, uin, col);
    e[2] = h3_extract_polycoeff(g, s, pco, col);
    e[3] = h3_regrid(g, s, col, n_grd, ug, xg, missing, regrid_method);
    e[4] = h3_prepare_remapping(g, r, xd, col);
    e[5] = h3_remap(g, s, r, ud, col);
    if (e[0] == 0) { n_act[col] = g.n_act[col]; m_act[col] = g.m_act[col]; }
  }
  // only columns whose call succeeded are handed back (the reference leaves the others untouched)
  std::vector<double> tmp((size_t)(p_ord + 1) * n * nc);
  from_dev(pco, tmp.data(), (p_ord + 1) * n_src, ncol);
  for (int c = 0; c < ncol; ++c)
    if (errs[6 * c + 2] == 0)
      memcpy(polycoeff + (size_t)c * (p_ord + 1) * n, tmp.data() + (size_t)c * (p_ord + 1) * n,
             sizeof(double) * (p_ord + 1) * n);
  tmp.assign(ng * nc, 0.0);
  from_dev(xg, tmp.data(), n_grd, ncol);
  for (int c = 0; c < ncol; ++c)
    if (errs[6 * c + 3] == 0) memcpy(x_grd + (size_t)c * ng, tmp.data() + (size_t)c * ng, sizeof(double) * ng);
  tmp.assign(nd * nc, 0.0);
  from_dev(ud, tmp.data(), n_dst, ncol);
  for (int c = 0; c < ncol; ++c)
    if (errs[6 * c + 5] == 0) memcpy(u_dst + (size_t)c * nd, tmp.data() + (size_t)c * nd, sizeof(double) * nd);
}
