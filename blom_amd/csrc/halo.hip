// xctilr (halo update) and xccrc (checksum) on device-resident planes.
//
// Single-tile form of xctilr (phy/mod_xc.F90:4374-4419): the reference fills the N/S halo
// of columns 1..ii first, then the E/W halo of rows 1-nhl..jj+nhl (corners come from the
// already updated N/S halo).  Both phases are data movement only, so one gather kernel
// computes for every halo point the interior source it ends up holding:
//   closed direction  -> vland (phy/mod_xc.F90:4382-4383, :4404-4405)
//   periodic direction-> wrapped index (phy/mod_xc.F90:4391-4392, :4413-4414)
// HBM traffic: (2*nhl*ii + 2*mhl*(jj+2*nhl)) * nlev * 16 B; negligible next to the stages.
#include "blomgpu_internal.h"

__global__ void k_xctilr_single(const DevView *Vp, double *__restrict__ a, int nlev, int mhl,
                                int nhl) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj;
  // enumerate halo targets: N/S strips (2*nhl rows x ii) then E/W strips (2*mhl cols x (jj+2nhl))
  const int nns = 2 * nhl * ii;
  const int new_ = 2 * mhl * (jj + 2 * nhl);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nns + new_) return;
  int i, j;
  if (t < nns) {
    int r = t / ii;                 // 0..2*nhl-1
    i = t % ii + 1;
    j = r < nhl ? -r : jj + (r - nhl) + 1;      // 0,-1,.. ; jj+1,..
  } else {
    t -= nns;
    int cidx = t % (2 * mhl);
    j = t / (2 * mhl) + 1 - nhl;
    i = cidx < mhl ? -cidx : ii + (cidx - mhl) + 1;
  }
  const bool inew = i < 1 || i > ii, ins = j < 1 || j > jj;
  const bool ew_closed = V.nreg == 0 || V.nreg == 4;
  const bool ns_closed = V.nreg <= 2;
  const bool land = (inew && ew_closed) || (ins && ns_closed);
  int is = i, js = j;
  if (i < 1) is = i + ii; else if (i > ii) is = i - ii;
  if (j < 1) js = j + jj; else if (j > jj) js = j - jj;
  const size_t dst = IDX(V, i, j), src = IDX(V, is, js);
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    double *pl = a + (size_t)k * V.nplane;
    pl[dst] = land ? V.P.vland : pl[src];
  }
}

int st_xctilr(blomgpu_ctx *c, double *base, int l1, int ld, int mh, int nh, int itype) {
  (void)itype;   // only distinguishes grids/vectors across the arctic seam (phy/mod_xc.F90:4248-4250)
  const DevView &h = c->h;
  if (h.nreg == 2) return ctx_fail(c, "xctilr: tripolar seam (nreg=2) not built yet");
  if (h.itdm != h.ii || h.jtdm != h.jj) return ctx_fail(c, "xctilr: multi-tile exchange not built yet");
  const int mhl = mh < 0 ? 0 : (mh > NBDY ? NBDY : mh);
  const int nhl = nh < 0 ? 0 : (nh > NBDY ? NBDY : nh);
  const int nlev = ld - l1 + 1;
  const int ntarget = 2 * nhl * h.ii + 2 * mhl * (h.jj + 2 * nhl);
  if (ntarget == 0 || nlev <= 0) return 0;
  dim3 grid((ntarget + 255) / 256, nlev > 64 ? 64 : nlev);
  hipLaunchKernelGGL(k_xctilr_single, grid, dim3(256), 0, c->stream, c->d,
                     base + (size_t)(l1 - 1) * h.nplane, nlev, mhl, nhl);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- xccrc (phy/mod_xc.F90:4164-4205, CRC-32 of phy/mod_crc32.F90) -----------------------
// Note: chksum() masks with ip/iu/iv/iq depending on itype; the C-ABI takes the field and
// derives nothing from itype, so the caller passes the mask through `mask_id` below.
__device__ inline unsigned crc_byte(unsigned crc, unsigned b) {
  unsigned k = (crc ^ b) & 255u;
#pragma unroll
  for (int j = 0; j < 8; j++) k = (k & 1u) ? (k >> 1) ^ 0xEDB88320u : (k >> 1);
  return (crc >> 8) ^ k;
}

__global__ void k_crc_strips(const DevView *Vp, const double *__restrict__ a, int nlev,
                             const int *__restrict__ mask, int nstrip, unsigned *__restrict__ out) {
  const DevView &V = *Vp;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nstrip * V.jj) return;
  const int j = t / nstrip + 1, s = t % nstrip;
  const int i1 = 1 + s * (2 * NBDY + 1);
  const int i2 = min(i1 + 2 * NBDY, V.ii);
  unsigned crc8p = 0;
  for (int i = i1; i <= i2; i++) {
    if (mask[IDX(V, i, j)] != 1) continue;
    unsigned crc = ~crc8p;                       // crc32(a(i,j,:), crc8p)
    for (int k = 0; k < nlev; k++) {
      unsigned long long bits = __double_as_longlong(a[(size_t)k * V.nplane + IDX(V, i, j)]);
      for (int b = 0; b < 8; b++) crc = crc_byte(crc, (unsigned)(bits >> (8 * b)) & 255u);
    }
    crc8p = ~crc;
  }
  out[t] = crc8p;
}

static unsigned host_crc_bytes(const unsigned char *p, size_t n, unsigned init) {
  unsigned crc = ~init;
  for (size_t x = 0; x < n; x++) {
    unsigned k = (crc ^ p[x]) & 255u;
    for (int j = 0; j < 8; j++) k = (k & 1u) ? (k >> 1) ^ 0xEDB88320u : (k >> 1);
    crc = (crc >> 8) ^ k;
  }
  return ~crc;
}

int st_crc(blomgpu_ctx *c, const double *base, int nlev, int itype, unsigned *crc) {
  const int g = itype % 10;   // phy/mod_checksum.F90:53-68
  const int *mask = g == 1 ? c->h.m[I_ip] : g == 2 ? c->h.m[I_iq] : g == 3 ? c->h.m[I_iu] : c->h.m[I_iv];
  const DevView &h = c->h;
  const int ns = (h.ii + 2 * NBDY + 1 - 1) / (2 * NBDY + 1);
  unsigned *dout = nullptr;
  HIPCHK(c, hipMalloc((void **)&dout, sizeof(unsigned) * ns * h.jj));
  hipLaunchKernelGGL(k_crc_strips, dim3((ns * h.jj + 63) / 64), dim3(64), 0, c->stream, c->d, base,
                     nlev, mask, ns, dout);
  std::vector<unsigned> hs((size_t)ns * h.jj);
  HIPCHK(c, hipMemcpyAsync(hs.data(), dout, sizeof(unsigned) * hs.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(dout);
  std::vector<unsigned> rows(h.jj);
  for (int j = 0; j < h.jj; j++) {
    unsigned crc8 = 0;
    for (int s = 0; s < ns; s++)
      crc8 = host_crc_bytes((const unsigned char *)&hs[(size_t)j * ns + s], 4, crc8);
    rows[j] = crc8;
  }
  *crc = host_crc_bytes((const unsigned char *)rows.data(), 4 * rows.size(), 0);
  return 0;
}
