// hor3map.hip -- batched HOR3MAP on MI355X: kernels + C ABI (include/blomgpu_hor3map.h).
// The per-column arithmetic is in hor3map_core.h; this file owns the HBM layout, the
// transposes between the caller's (level, column) arrays and the [level][column] device layout,
// the launches and the error plumbing.  No host fallback: without a HIP device grid_create fails.
#include "hor3map_core.h"
#include "hor3map_pqm.h"
#include "hor3map_ppm_fused.h"
#include "../../include/blomgpu_hor3map.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int H3_BLOCK = 64;          // one wavefront per block: ~1.7k blocks for a 106k-column slab
constexpr unsigned long long H3_NOFAIL = ~0ull;

struct Pool {
  std::vector<void *> ptrs;
  bool ok = true;
  template <class T> T *get(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) ? n * sizeof(T) : sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
    ptrs.push_back(p);
    return (T *)p;
  }
  void release() {
    for (void *p : ptrs) (void)hipFree(p);
    ptrs.clear();
  }
};

}  // namespace

struct blomgpu_h3m_grid {
  H3Grid g{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  Pool pool;
  bool device_io = false, check = true;
  bool sticky = false;        // the first failing column is kept across calls until h3m_first_error reads it (one read-back per sequence)
  bool plane_io = false;      // device pointers in the library's own [level][column] layout: no transpose (the model's planes)
  bool own_stream = true;
  unsigned long long *first_fail = nullptr;   // (column << 8 | errstat) of the first failing column
  double *stage_in = nullptr, *stage_out = nullptr;   // caller-layout staging for host pointers
  size_t stage_in_n = 0, stage_out_n = 0;
  double *tin = nullptr, *tout = nullptr;             // [level][column] staging
  size_t tin_n = 0, tout_n = 0;
  double *tmany = nullptr;                            // [field][level][column] staging of the *_many entries
  size_t tmany_n = 0;
  int *err_many = nullptr;                            // per-field, per-column status of the last *_many call
  std::vector<blomgpu_h3m_src *> srcs;
  std::vector<blomgpu_h3m_map *> maps;
};
struct blomgpu_h3m_src {
  H3Src s{};
  blomgpu_h3m_grid *grid = nullptr;
  Pool pool;
};
struct blomgpu_h3m_map {
  H3Map r{};
  blomgpu_h3m_grid *grid = nullptr;
  Pool pool;
};

namespace {

// caller layout a(m, ncol) (level fastest) <-> device layout [level][column], through a padded LDS tile
__global__ __launch_bounds__(256) void k_h3m_to_device(const double *__restrict__ a, double *__restrict__ t,
                                                       int m, int nc) {
  __shared__ double tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int cc = ty; cc < 32; cc += 8) {
    const int c = c0 + cc, r = r0 + tx;
    if (c < nc && r < m) tile[cc][tx] = a[(size_t)c * m + r];
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    if (c < nc && r < m) t[(size_t)r * nc + c] = tile[tx][rr];
  }
}
__global__ __launch_bounds__(256) void k_h3m_from_device(const double *__restrict__ t, double *__restrict__ a,
                                                         int m, int nc) {
  __shared__ double tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, c = c0 + tx;
    if (c < nc && r < m) tile[rr][tx] = t[(size_t)r * nc + c];
  }
  __syncthreads();
  for (int cc = ty; cc < 32; cc += 8) {
    const int c = c0 + cc, r = r0 + tx;
    if (c < nc && r < m) a[(size_t)c * m + r] = tile[tx][cc];
  }
}

__device__ inline void h3_report(int *err, unsigned long long *first_fail, int col, int e) {
  err[col] = e;
  if (e != H3_NOERR) atomicMin(first_fail, ((unsigned long long)col << 8) | (unsigned)e);
}

__global__ __launch_bounds__(H3_BLOCK) void k_h3m_prepare(H3Grid g, unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  h3_report(g.err, ff, col, h3_prepare_reconstruction(g, col));
}
__global__ __launch_bounds__(H3_BLOCK) void k_h3m_reconstruct(H3Grid g, H3Src s, const double *uin,
                                                              unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  h3_report(g.err, ff, col, h3_reconstruct(g, s, uin, col));
}
__global__ __launch_bounds__(H3_BLOCK) void k_h3m_extract(H3Grid g, H3Src s, double *out, unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  h3_report(g.err, ff, col, h3_extract_polycoeff(g, s, out, col));
}
__global__ __launch_bounds__(H3_BLOCK) void k_h3m_regrid(H3Grid g, H3Src s, int ng, const double *ugrd,
                                                         double *xgrd, double missing, int method,
                                                         unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  h3_report(g.err, ff, col, h3_regrid(g, s, col, ng, ugrd, xgrd, missing, method));
}
__global__ __launch_bounds__(H3_BLOCK) void k_h3m_prepare_remap(H3Grid g, H3Map r, const double *xdst,
                                                                unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  h3_report(g.err, ff, col, h3_prepare_remapping(g, r, xdst, col));
}
__global__ __launch_bounds__(H3_BLOCK) void k_h3m_remap(H3Grid g, H3Src s, H3Map r, double *udst,
                                                        unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  h3_report(g.err, ff, col, h3_remap(g, s, r, udst, col));
}

// Several source fields on one grid in a single launch (blockIdx.y = field): the column routines are
// bound by memory latency at 6.5 wavefronts per CU (measured: time grows 1.8x from 26k to 106k columns),
// and BLOM's callers have ntr tracers per grid, so the fields of a tracer loop fill the machine.
#define H3_MAXF 8
struct H3SrcSet {
  H3Src s[H3_MAXF];
  double *io[H3_MAXF];          // [level][column] staging of each field's caller array
};
__global__ __launch_bounds__(H3_BLOCK) void k_h3m_reconstruct_many(H3Grid g, H3SrcSet S, int *err_f,
                                                                   unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  const int f = blockIdx.y;
  h3_report(err_f + (size_t)f * g.nc, ff, col, h3_reconstruct(g, S.s[f], S.io[f], col));
}
__global__ __launch_bounds__(H3_BLOCK) void k_h3m_remap_many(H3Grid g, H3SrcSet S, H3Map r, int *err_f,
                                                             unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  const int f = blockIdx.y;
  h3_report(err_f + (size_t)f * g.nc, ff, col, h3_remap(g, S.s[f], r, S.io[f], col));
}

__global__ __launch_bounds__(H3_BLOCK) void k_h3m_extract_many(H3Grid g, H3SrcSet S, int *err_f, unsigned long long *ff) {
  const int col = blockIdx.x * H3_BLOCK + threadIdx.x;
  if (col >= g.nc || (g.active && !g.active[col])) return;
  const int f = blockIdx.y;
  h3_report(err_f + (size_t)f * g.nc, ff, col, h3_extract_polycoeff(g, S.s[f], S.io[f], col));
}

constexpr int E_DEVICE = -1, E_ALLOC = -2, E_HANDLE = -3, E_ARG = -4;
thread_local std::string g_devmsg;

int dev_fail(int code, const char *what, hipError_t e) {
  g_devmsg = std::string("blomgpu_h3m: ") + what + ": " + hipGetErrorString(e);
  return code;
}
#define H3CHK(call)                                              \
  do {                                                           \
    hipError_t e_ = (call);                                      \
    if (e_ != hipSuccess) return dev_fail(E_DEVICE, #call, e_);  \
  } while (0)

bool grow(blomgpu_h3m_grid *G, double *&p, size_t &have, size_t need) {
  if (need <= have) return true;
  if (p) (void)hipFree(p);
  p = nullptr;
  have = 0;
  if (hipMalloc((void **)&p, need * sizeof(double)) != hipSuccess) return false;
  have = need;
  return true;
}

// bring a caller array a(m, ncol) into G->tin as [m][ncol]
int load_input(blomgpu_h3m_grid *G, const double *a, int m, double *dst) {
  const int nc = G->g.nc;
  const size_t n = (size_t)m * nc;
  const double *src = a;
  if (G->plane_io) {
    H3CHK(hipMemcpyAsync(dst, a, n * sizeof(double), hipMemcpyDeviceToDevice, G->stream));
    return 0;
  }
  if (!G->device_io) {
    if (!grow(G, G->stage_in, G->stage_in_n, n)) return E_ALLOC;
    H3CHK(hipMemcpyAsync(G->stage_in, a, n * sizeof(double), hipMemcpyHostToDevice, G->stream));
    src = G->stage_in;
  }
  dim3 grid((nc + 31) / 32, (m + 31) / 32);
  hipLaunchKernelGGL(k_h3m_to_device, grid, dim3(256), 0, G->stream, src, dst, m, nc);
  return 0;
}
int store_output(blomgpu_h3m_grid *G, const double *t, int m, double *a) {
  const int nc = G->g.nc;
  const size_t n = (size_t)m * nc;
  double *dst = a;
  if (G->plane_io) {
    H3CHK(hipMemcpyAsync(a, t, n * sizeof(double), hipMemcpyDeviceToDevice, G->stream));
    return 0;
  }
  if (!G->device_io) {
    if (!grow(G, G->stage_out, G->stage_out_n, n)) return E_ALLOC;
    dst = G->stage_out;
  }
  dim3 grid((nc + 31) / 32, (m + 31) / 32);
  hipLaunchKernelGGL(k_h3m_from_device, grid, dim3(256), 0, G->stream, t, dst, m, nc);
  if (!G->device_io) {
    H3CHK(hipMemcpyAsync(a, dst, n * sizeof(double), hipMemcpyDeviceToHost, G->stream));
    H3CHK(hipStreamSynchronize(G->stream));
  }
  return 0;
}
int begin_call(blomgpu_h3m_grid *G) {
  H3CHK(hipSetDevice(G->device));
  if (!G->sticky) H3CHK(hipMemsetAsync(G->first_fail, 0xFF, sizeof(unsigned long long), G->stream));
  return 0;
}
int end_call(blomgpu_h3m_grid *G) {
  H3CHK(hipGetLastError());
  if (!G->check) return 0;
  unsigned long long ff = H3_NOFAIL;
  H3CHK(hipMemcpyAsync(&ff, G->first_fail, sizeof(ff), hipMemcpyDeviceToHost, G->stream));
  H3CHK(hipStreamSynchronize(G->stream));
  return ff == H3_NOFAIL ? 0 : (int)(ff & 0xFF);
}
inline dim3 col_grid(const blomgpu_h3m_grid *G) { return dim3((G->g.nc + H3_BLOCK - 1) / H3_BLOCK); }

}  // namespace

extern "C" {

int blomgpu_h3m_grid_create(blomgpu_h3m_grid **out, int device, int ncol, int n_src, int method,
                            int left_bndr_ord, int right_bndr_ord) {
  if (!out || ncol < 1 || n_src < 1) { g_devmsg = "blomgpu_h3m_grid_create: bad argument"; return E_ARG; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_devmsg = "blomgpu_h3m_grid_create: no HIP device (there is no host fallback)";
    return E_DEVICE;
  }
  // initialize_rcgs (mod_hor3map.F90:3620-3656): polynomial order and clamping of the boundary orders
  int p_ord, lbo = left_bndr_ord, rbo = right_bndr_ord;
  switch (method) {
    case H3_PCM: p_ord = 0; break;
    case H3_PLM: p_ord = 1; break;
    case H3_PPM:
      p_ord = 2;
      lbo = lbo == 0 ? H3_EB_MAX_PPM : (lbo < 1 ? 1 : (lbo > H3_EB_MAX_PPM ? H3_EB_MAX_PPM : lbo));
      rbo = rbo == 0 ? H3_EB_MAX_PPM : (rbo < 1 ? 1 : (rbo > H3_EB_MAX_PPM ? H3_EB_MAX_PPM : rbo));
      break;
    case H3_PQM:
      p_ord = 4;
      lbo = lbo == 0 ? H3_EB_MAX_PQM : (lbo < 1 ? 1 : (lbo > H3_EB_MAX_PQM ? H3_EB_MAX_PQM : lbo));
      rbo = rbo == 0 ? H3_EB_MAX_PQM : (rbo < 1 ? 1 : (rbo > H3_EB_MAX_PQM ? H3_EB_MAX_PQM : rbo));
      break;
    default: return H3_INVALID_RECON_METHOD;
  }
  H3CHK(hipSetDevice(device));
  auto *G = new blomgpu_h3m_grid;
  G->device = device;
  H3Grid &g = G->g;
  g.nc = ncol; g.n_src = n_src; g.method = method; g.left_bndr_ord = lbo; g.right_bndr_ord = rbo;
  g.p_ord = p_ord; g.ncoef = p_ord + 2;
  const size_t nc = ncol, n = n_src;
  Pool &P = G->pool;
  g.xin = P.get<double>((n + 1) * nc);
  g.x_eps = P.get<double>(nc);
  g.x_edge = P.get<double>((n + 1) * nc);
  g.h = P.get<double>(n * nc);
  g.hi = P.get<double>(n * nc);
  g.hci = P.get<double>(n * nc);
  g.w = P.get<double>(n * nc);
  g.tde = P.get<double>(g.ncoef * n * nc);
  g.tds = P.get<double>(method == H3_PQM ? g.ncoef * n * nc : 1);
  g.lblu = P.get<double>((size_t)H3_LD * H3_LD * nc);
  g.rblu = P.get<double>((size_t)H3_LD * H3_LD * nc);
  g.sdi = P.get<int>(n * nc);
  g.n_act = P.get<int>(nc); g.m_act = P.get<int>(nc); g.lb_act = P.get<int>(nc); g.rb_act = P.get<int>(nc);
  g.prepared = P.get<int>(nc); g.err = P.get<int>(nc);
  g.prev = P.get<int>(n * nc); g.next = P.get<int>(n * nc);
  G->first_fail = P.get<unsigned long long>(1);
  if (!P.ok) {
    P.release();
    delete G;
    g_devmsg = "blomgpu_h3m_grid_create: hipMalloc failed";
    return E_ALLOC;
  }
  H3CHK(hipStreamCreate(&G->stream));
  H3CHK(hipEventCreate(&G->ev0));
  H3CHK(hipEventCreate(&G->ev1));
  H3CHK(hipMemsetAsync(g.prepared, 0, nc * sizeof(int), G->stream));
  H3CHK(hipMemsetAsync(g.err, 0, nc * sizeof(int), G->stream));
  H3CHK(hipMemsetAsync(g.n_act, 0, nc * sizeof(int), G->stream));
  H3CHK(hipMemsetAsync(g.m_act, 0, nc * sizeof(int), G->stream));
  H3CHK(hipMemsetAsync(G->first_fail, 0xFF, sizeof(unsigned long long), G->stream));     // "no column has failed"
  // boundary LU planes are read only where a prepare wrote them, except in the reference's own
  // corner cases (:1734); keep them defined
  H3CHK(hipMemsetAsync(g.lblu, 0, (size_t)H3_LD * H3_LD * nc * sizeof(double), G->stream));
  H3CHK(hipMemsetAsync(g.rblu, 0, (size_t)H3_LD * H3_LD * nc * sizeof(double), G->stream));
  H3CHK(hipStreamSynchronize(G->stream));
  *out = G;
  return 0;
}

int blomgpu_h3m_src_create(blomgpu_h3m_grid *G, blomgpu_h3m_src **out, int limiting, int pc_left_bndr,
                           int pc_right_bndr) {
  if (!G || !out) return E_HANDLE;
  H3CHK(hipSetDevice(G->device));
  auto *S = new blomgpu_h3m_src;
  S->grid = G;
  H3Src &s = S->s;
  s.limiting = limiting; s.pc_left = pc_left_bndr != 0; s.pc_right = pc_right_bndr != 0;
  const size_t nc = G->g.nc, n = G->g.n_src;
  Pool &P = S->pool;
  s.u = P.get<double>(n * nc); s.uel = P.get<double>(n * nc); s.uer = P.get<double>(n * nc);
  const bool pqm = G->g.method == H3_PQM;
  s.usl = P.get<double>(pqm ? n * nc : 1); s.usr = P.get<double>(pqm ? n * nc : 1);
  s.pc = P.get<double>((size_t)(G->g.p_ord + 1) * n * nc);
  s.u_range = P.get<double>(nc); s.u_eps = P.get<double>(nc); s.uu_eps = P.get<double>(nc);
  s.reconstructed = P.get<int>(nc);
  s.err = G->g.err;
  s.wk = P.get<double>(4 * (n + 1) * nc);
  if (!P.ok) {
    P.release();
    delete S;
    g_devmsg = "blomgpu_h3m_src_create: hipMalloc failed";
    return H3_FAILED_TO_ALLOCATE_RCSS;
  }
  H3CHK(hipMemsetAsync(s.reconstructed, 0, nc * sizeof(int), G->stream));
  H3CHK(hipStreamSynchronize(G->stream));
  G->srcs.push_back(S);
  *out = S;
  return 0;
}

int blomgpu_h3m_map_create(blomgpu_h3m_grid *G, blomgpu_h3m_map **out, int n_dst) {
  if (!G || !out || n_dst < 1) return E_HANDLE;
  H3CHK(hipSetDevice(G->device));
  auto *M = new blomgpu_h3m_map;
  M->grid = G;
  H3Map &r = M->r;
  r.n_dst = n_dst;
  const size_t nc = G->g.nc, n = G->g.n_src, nd = n_dst;
  Pool &P = M->pool;
  r.lim = P.get<double>((n + nd) * nc); r.wgt = P.get<double>((n + nd) * nc);
  r.nseg = P.get<int>(n * nc); r.sdst = P.get<int>((n + nd) * nc);
  r.prepared = P.get<int>(nc);
  r.err = G->g.err;
  r.hdst = P.get<double>(nd * nc);
  if (!P.ok) {
    P.release();
    delete M;
    g_devmsg = "blomgpu_h3m_map_create: hipMalloc failed";
    return H3_FAILED_TO_ALLOCATE_RMS;
  }
  H3CHK(hipMemsetAsync(r.prepared, 0, nc * sizeof(int), G->stream));
  H3CHK(hipStreamSynchronize(G->stream));
  G->maps.push_back(M);
  *out = M;
  return 0;
}

static void unlink_src(blomgpu_h3m_grid *G, blomgpu_h3m_src *S) {
  for (size_t i = 0; i < G->srcs.size(); ++i)
    if (G->srcs[i] == S) { G->srcs.erase(G->srcs.begin() + i); break; }
}
static void unlink_map(blomgpu_h3m_grid *G, blomgpu_h3m_map *M) {
  for (size_t i = 0; i < G->maps.size(); ++i)
    if (G->maps[i] == M) { G->maps.erase(G->maps.begin() + i); break; }
}
void blomgpu_h3m_src_free(blomgpu_h3m_src *S) {
  if (!S) return;
  if (S->grid) { (void)hipStreamSynchronize(S->grid->stream); unlink_src(S->grid, S); }
  S->pool.release();
  delete S;
}
void blomgpu_h3m_map_free(blomgpu_h3m_map *M) {
  if (!M) return;
  if (M->grid) { (void)hipStreamSynchronize(M->grid->stream); unlink_map(M->grid, M); }
  M->pool.release();
  delete M;
}
void blomgpu_h3m_grid_free(blomgpu_h3m_grid *G) {
  if (!G) return;
  (void)hipSetDevice(G->device);
  (void)hipStreamSynchronize(G->stream);
  while (!G->srcs.empty()) blomgpu_h3m_src_free(G->srcs.back());
  while (!G->maps.empty()) blomgpu_h3m_map_free(G->maps.back());
  G->pool.release();
  if (G->err_many) (void)hipFree(G->err_many);
  for (double *p : {G->stage_in, G->stage_out, G->tin, G->tout, G->tmany})
    if (p) (void)hipFree(p);
  if (G->ev0) (void)hipEventDestroy(G->ev0);
  if (G->ev1) (void)hipEventDestroy(G->ev1);
  if (G->stream && G->own_stream) (void)hipStreamDestroy(G->stream);
  delete G;
}

int blomgpu_h3m_set_io(blomgpu_h3m_grid *G, int device_pointers, int check_errors) {
  if (!G) return E_HANDLE;
  G->device_io = device_pointers != 0;
  G->plane_io = device_pointers == 2;
  G->check = check_errors != 0;
  return 0;
}

}  // extern "C"
// inside the library (stage_ale.hip): a sequence of calls whose status is read back once.  h3m_sequence_begin clears the record
// and makes it sticky (calls return 0 unless the device layer fails); h3m_sequence_end returns an errstat of the lowest column
// that failed in any call of the sequence (the kernels record failures with atomicMin on column << 8 | errstat) -- enough to
// stop the stage with the reference's message for that column.
int h3m_sequence_begin(blomgpu_h3m_grid *G) {
  if (!G) return E_HANDLE;
  H3CHK(hipMemsetAsync(G->first_fail, 0xFF, sizeof(unsigned long long), G->stream));
  G->sticky = true;
  G->check = false;
  return 0;
}
// Deferred form, inside blomgpu_step: the record is neither cleared at the start of a sequence nor read back at its end (no
// host synchronisation: the step's launches can be captured into a graph); h3m_sequence_poll reads it -- the model context
// does so with its other sticky error words, every check_period steps -- and clears it.
int h3m_sequence_begin_deferred(blomgpu_h3m_grid *G) {
  if (!G) return E_HANDLE;
  G->sticky = true;
  G->check = false;
  return 0;
}
int h3m_sequence_end_deferred(blomgpu_h3m_grid *G) {
  if (!G) return E_HANDLE;
  G->sticky = false;
  G->check = true;
  return 0;
}
int h3m_sequence_poll(blomgpu_h3m_grid *G, unsigned long long *column) {
  if (!G) return E_HANDLE;
  unsigned long long ff = H3_NOFAIL;
  H3CHK(hipMemcpyAsync(&ff, G->first_fail, sizeof(ff), hipMemcpyDeviceToHost, G->stream));
  H3CHK(hipStreamSynchronize(G->stream));
  if (ff == H3_NOFAIL) return 0;
  H3CHK(hipMemsetAsync(G->first_fail, 0xFF, sizeof(unsigned long long), G->stream));
  if (column) *column = ff >> 8;
  return (int)(ff & 0xFF);
}
int h3m_sequence_end(blomgpu_h3m_grid *G) {
  if (!G) return E_HANDLE;
  G->sticky = false;
  G->check = true;
  unsigned long long ff = H3_NOFAIL;
  H3CHK(hipMemcpyAsync(&ff, G->first_fail, sizeof(ff), hipMemcpyDeviceToHost, G->stream));
  H3CHK(hipStreamSynchronize(G->stream));
  return ff == H3_NOFAIL ? 0 : (int)(ff & 0xFF);
}
// inside the library (stage_ale.hip): the grid works on the model context's stream
int h3m_use_stream(blomgpu_h3m_grid *G, hipStream_t stream) {
  if (!G) return E_HANDLE;
  H3CHK(hipStreamSynchronize(G->stream));
  if (G->own_stream && G->stream) (void)hipStreamDestroy(G->stream);
  G->stream = stream;
  G->own_stream = false;
  return 0;
}
// inside the library: the launches that follow go to another stream (the caller orders the streams with events); no
// synchronisation, the grid does not own either stream
int h3m_set_stream(blomgpu_h3m_grid *G, hipStream_t stream) {
  if (!G || G->own_stream) return E_HANDLE;
  G->stream = stream;
  return 0;
}
// inside the library (stage_ale.hip): columns whose entry in `active` (one int per column, device memory) is 0 take no part in
// the launches that follow -- land and the halo beyond the ring the stage needs; their structures and outputs stay as they are
int h3m_set_active(blomgpu_h3m_grid *G, const int *active) {
  if (!G) return E_HANDLE;
  G->g.active = active;
  return 0;
}
// inside the library: the polynomial coefficients of several sources of one grid in one launch, written where the caller's
// [coefficient * level][column] planes lie
int h3m_extract_polycoeff_many(blomgpu_h3m_grid *G, int nf, blomgpu_h3m_src *const *srcs, double *const *outs) {
  if (!G || !G->plane_io || nf < 1 || nf > H3_MAXF) return E_HANDLE;
  if (!G->err_many && hipMalloc((void **)&G->err_many, sizeof(int) * H3_MAXF * G->g.nc) != hipSuccess) return E_ALLOC;
  H3SrcSet S{};
  for (int f = 0; f < nf; f++) {
    if (!srcs[f] || srcs[f]->grid != G) return H3_INCONSISTENT_RCGS;
    S.s[f] = srcs[f]->s;
    S.io[f] = outs[f];
  }
  hipLaunchKernelGGL(k_h3m_extract_many, dim3(col_grid(G).x, nf), dim3(H3_BLOCK), 0, G->stream, G->g, S, G->err_many, G->first_fail);
  return end_call(G);
}
extern "C" {

int blomgpu_h3m_prepare_reconstruction(blomgpu_h3m_grid *G, const double *x_edge_src) {
  if (!G || !x_edge_src) return E_HANDLE;
  int rc = begin_call(G);
  if (rc) return rc;
  // the caller's edges are read by this kernel alone (hor3map_core.h: g.xin): planes are taken where they lie
  H3Grid g = G->g;
  if (G->plane_io) g.xin = const_cast<double *>(x_edge_src);
  else if ((rc = load_input(G, x_edge_src, G->g.n_src + 1, G->g.xin))) return rc;
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_prepare, col_grid(G), dim3(H3_BLOCK), 0, G->stream, g, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  return end_call(G);
}

int blomgpu_h3m_reconstruct(blomgpu_h3m_grid *G, blomgpu_h3m_src *S, const double *u_src) {
  if (!G || !S || !u_src) return E_HANDLE;
  if (S->grid != G) return H3_INCONSISTENT_RCGS;
  int rc = begin_call(G);
  if (rc) return rc;
  const size_t n = (size_t)G->g.n_src * G->g.nc;
  const double *uin = u_src;
  if (!G->plane_io) {
    if (!grow(G, G->tin, G->tin_n, n)) return E_ALLOC;
    if ((rc = load_input(G, u_src, G->g.n_src, G->tin))) return rc;
    uin = G->tin;
  }
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_reconstruct, col_grid(G), dim3(H3_BLOCK), 0, G->stream, G->g, S->s, uin, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  return end_call(G);
}

int blomgpu_h3m_extract_polycoeff(blomgpu_h3m_src *S, double *polycoeff) {
  if (!S || !polycoeff) return E_HANDLE;
  blomgpu_h3m_grid *G = S->grid;
  int rc = begin_call(G);
  if (rc) return rc;
  const int m = (G->g.p_ord + 1) * G->g.n_src;
  double *out = polycoeff;
  if (!G->plane_io) {
    if (!grow(G, G->tout, G->tout_n, (size_t)m * G->g.nc)) return E_ALLOC;
    out = G->tout;
  }
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_extract, col_grid(G), dim3(H3_BLOCK), 0, G->stream, G->g, S->s, out, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  if (!G->plane_io && (rc = store_output(G, G->tout, m, polycoeff))) return rc;
  return end_call(G);
}

int blomgpu_h3m_regrid(blomgpu_h3m_src *S, int n_grd, const double *u_edge_grd, double *x_edge_grd,
                       double missing_value, int regrid_method) {
  if (!S || !u_edge_grd || !x_edge_grd || n_grd < 1) return E_HANDLE;
  if (regrid_method == 0) regrid_method = H3_REGRID_METHOD_1;      // absent optional argument (:4502)
  if (regrid_method != H3_REGRID_METHOD_1 && regrid_method != H3_REGRID_METHOD_2) return H3_INVALID_REGRID_METHOD;
  blomgpu_h3m_grid *G = S->grid;
  int rc = begin_call(G);
  if (rc) return rc;
  const size_t n = (size_t)n_grd * G->g.nc;
  const double *ugrd = u_edge_grd;
  double *xgrd = x_edge_grd;
  if (!G->plane_io) {
    if (!grow(G, G->tin, G->tin_n, n) || !grow(G, G->tout, G->tout_n, n)) return E_ALLOC;
    if ((rc = load_input(G, u_edge_grd, n_grd, G->tin))) return rc;
    ugrd = G->tin;
    xgrd = G->tout;
  }
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_regrid, col_grid(G), dim3(H3_BLOCK), 0, G->stream, G->g, S->s, n_grd, ugrd, xgrd, missing_value,
                     regrid_method, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  if (!G->plane_io && (rc = store_output(G, G->tout, n_grd, x_edge_grd))) return rc;
  return end_call(G);
}

int blomgpu_h3m_prepare_remapping(blomgpu_h3m_grid *G, blomgpu_h3m_map *M, const double *x_edge_dst) {
  if (!G || !M || !x_edge_dst) return E_HANDLE;
  if (M->grid != G) return H3_INCONSISTENT_RCGS;
  int rc = begin_call(G);
  if (rc) return rc;
  const int m = M->r.n_dst + 1;
  const double *xdst = x_edge_dst;
  if (!G->plane_io) {
    if (!grow(G, G->tin, G->tin_n, (size_t)m * G->g.nc)) return E_ALLOC;
    if ((rc = load_input(G, x_edge_dst, m, G->tin))) return rc;
    xdst = G->tin;
  }
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_prepare_remap, col_grid(G), dim3(H3_BLOCK), 0, G->stream, G->g, M->r, xdst, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  return end_call(G);
}

int blomgpu_h3m_remap(blomgpu_h3m_src *S, blomgpu_h3m_map *M, double *u_dst) {
  if (!S || !M || !u_dst) return E_HANDLE;
  if (S->grid != M->grid) return H3_INCONSISTENT_RCGS;
  blomgpu_h3m_grid *G = S->grid;
  int rc = begin_call(G);
  if (rc) return rc;
  const int m = M->r.n_dst;
  double *out = u_dst;
  if (!G->plane_io) {
    if (!grow(G, G->tout, G->tout_n, (size_t)m * G->g.nc)) return E_ALLOC;
    out = G->tout;
  }
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_remap, col_grid(G), dim3(H3_BLOCK), 0, G->stream, G->g, S->s, M->r, out, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  if (!G->plane_io && (rc = store_output(G, G->tout, m, u_dst))) return rc;
  return end_call(G);
}

int blomgpu_h3m_reconstruct_many(blomgpu_h3m_grid *G, int nf, blomgpu_h3m_src *const *srcs,
                                 const double *const *u_srcs) {
  if (!G || !srcs || !u_srcs || nf < 1 || nf > H3_MAXF) return E_HANDLE;
  int rc = begin_call(G);
  if (rc) return rc;
  const size_t per = (size_t)G->g.n_src * G->g.nc;
  if (!G->plane_io && !grow(G, G->tmany, G->tmany_n, per * H3_MAXF)) return E_ALLOC;
  if (!G->err_many && hipMalloc((void **)&G->err_many, sizeof(int) * H3_MAXF * G->g.nc) != hipSuccess) return E_ALLOC;
  H3SrcSet S{};
  for (int f = 0; f < nf; f++) {
    if (!srcs[f] || srcs[f]->grid != G) return H3_INCONSISTENT_RCGS;
    S.s[f] = srcs[f]->s;
    if (G->plane_io) { S.io[f] = const_cast<double *>(u_srcs[f]); continue; }
    S.io[f] = G->tmany + (size_t)f * per;
    if ((rc = load_input(G, u_srcs[f], G->g.n_src, S.io[f]))) return rc;
  }
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_reconstruct_many, dim3(col_grid(G).x, nf), dim3(H3_BLOCK), 0, G->stream, G->g, S,
                     G->err_many, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  return end_call(G);
}

int blomgpu_h3m_remap_many(int nf, blomgpu_h3m_src *const *srcs, blomgpu_h3m_map *M, double *const *u_dsts) {
  if (!M || !srcs || !u_dsts || nf < 1 || nf > H3_MAXF) return E_HANDLE;
  blomgpu_h3m_grid *G = M->grid;
  int rc = begin_call(G);
  if (rc) return rc;
  const size_t per = (size_t)M->r.n_dst * G->g.nc;
  if (!G->plane_io && !grow(G, G->tmany, G->tmany_n, per * H3_MAXF)) return E_ALLOC;
  if (!G->err_many && hipMalloc((void **)&G->err_many, sizeof(int) * H3_MAXF * G->g.nc) != hipSuccess) return E_ALLOC;
  H3SrcSet S{};
  for (int f = 0; f < nf; f++) {
    if (!srcs[f] || srcs[f]->grid != G) return H3_INCONSISTENT_RCGS;
    S.s[f] = srcs[f]->s;
    S.io[f] = G->plane_io ? u_dsts[f] : G->tmany + (size_t)f * per;
  }
  H3CHK(hipEventRecord(G->ev0, G->stream));
  hipLaunchKernelGGL(k_h3m_remap_many, dim3(col_grid(G).x, nf), dim3(H3_BLOCK), 0, G->stream, G->g, S, M->r,
                     G->err_many, G->first_fail);
  H3CHK(hipEventRecord(G->ev1, G->stream));
  for (int f = 0; f < nf && !G->plane_io; f++)
    if ((rc = store_output(G, S.io[f], M->r.n_dst, u_dsts[f]))) return rc;
  return end_call(G);
}

int blomgpu_h3m_errstat(blomgpu_h3m_grid *G, int *errstat) {
  if (!G || !errstat) return E_HANDLE;
  H3CHK(hipSetDevice(G->device));
  H3CHK(hipStreamSynchronize(G->stream));
  H3CHK(hipMemcpy(errstat, G->g.err, (size_t)G->g.nc * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}
int blomgpu_h3m_grid_info(blomgpu_h3m_grid *G, int *n_src_actual, int *method_actual) {
  if (!G) return E_HANDLE;
  H3CHK(hipSetDevice(G->device));
  H3CHK(hipStreamSynchronize(G->stream));
  if (n_src_actual)
    H3CHK(hipMemcpy(n_src_actual, G->g.n_act, (size_t)G->g.nc * sizeof(int), hipMemcpyDeviceToHost));
  if (method_actual)
    H3CHK(hipMemcpy(method_actual, G->g.m_act, (size_t)G->g.nc * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}
int blomgpu_h3m_sync(blomgpu_h3m_grid *G) {
  if (!G) return E_HANDLE;
  H3CHK(hipSetDevice(G->device));
  H3CHK(hipStreamSynchronize(G->stream));
  return 0;
}
int blomgpu_h3m_last_kernel_ms(blomgpu_h3m_grid *G, float *ms) {
  if (!G || !ms) return E_HANDLE;
  H3CHK(hipEventSynchronize(G->ev1));
  H3CHK(hipEventElapsedTime(ms, G->ev0, G->ev1));
  return 0;
}

const char *blomgpu_h3m_errstr(int e) {
  // the reference's messages (mod_hor3map.F90:84-106)
  static const char *msg[] = {
      "",
      "Invalid reconstruction method!",
      "Cannot resize initialized reconstruction grid data structure!",
      "Source grid edges do not monotonically increase or decrease!",
      "Source grid extent too small!",
      "Failed to allocate reconstruction grid data structure!",
      "Call 'prepare_reconstruction' first!",
      "Cannot resize initialized remapping data structure!",
      "Inconsistent source and destination grid range!",
      "Destination grid edges do not monotonically increase or decrease!",
      "Failed to allocate remapping data structure!",
      "Size mismatch between source grid edges and data array!",
      "Failed to allocate reconstruction source data structure!",
      "Invalid limiting method for PLM!",
      "Invalid limiting method for PPM!",
      "Invalid limiting method for PQM!",
      "Call 'reconstruct' first!",
      "Invalid regrid method!",
      "Size mismatch between grid edge values and locations!",
      "Call 'prepare_remapping' first!",
      "Size mismatch between destination grid edges and data array!",
      "Array index of data structure is out of bounds!",
      "Inconsistent data structure for reconstruction and remapping!"};
  if (e >= 1 && e <= 22) return msg[e];
  if (e == 0) return "";
  return g_devmsg.empty() ? "blomgpu_h3m: device layer failure" : g_devmsg.c_str();
}

}  // extern "C"
