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

// One gather kernel serves a single tile and the in-process multi-tile transport: src[dy+1][dx+1]
// is the same (field, level) plane stack in the tile that owns the halo strip in direction
// (dx,dy) -- the tile itself where the direction is periodic and the grid has one tile there,
// nullptr where the domain is closed (vland).  Halo width <= nbdy < tile extent, so a halo point
// has exactly one owner.
struct HaloSrc {
  const double *p[3][3];
};

__global__ void k_xctilr_gather(const DevView *__restrict__ Vp, double *__restrict__ a, HaloSrc S, int nlev, int mhl, int nhl) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj;
  // enumerate halo targets: N/S strips (2*nhl rows x ii) then E/W strips (2*mhl cols x (jj+2nhl))
  const int nns = 2 * nhl * ii;
  const int new_ = 2 * mhl * (jj + 2 * nhl);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nns + new_) return;
  int i, j;
  if (t < nns) {
    const int r = t / ii;                 // 0..2*nhl-1
    i = t % ii + 1;
    j = r < nhl ? -r : jj + (r - nhl) + 1;      // 0,-1,.. ; jj+1,..
  } else {
    t -= nns;
    const int cidx = t % (2 * mhl);
    j = t / (2 * mhl) + 1 - nhl;
    i = cidx < mhl ? -cidx : ii + (cidx - mhl) + 1;
  }
  const int dx = i < 1 ? -1 : (i > ii ? 1 : 0), dy = j < 1 ? -1 : (j > jj ? 1 : 0);
  const double *sp = S.p[dy + 1][dx + 1];
  const size_t dst = IDX(V, i, j), src = IDX(V, i - dx * ii, j - dy * jj);
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    a[dst + o] = sp ? sp[src + o] : V.P.vland;
  }
}

// Arctic patch (nreg = 2, one tile), phy/mod_xc.F90:4262-4372: closed in the south, periodic in i,
// and across the last row the grid folds onto itself -- row jj+j takes the mirrored row jj-1-j (p-,
// u-grid, from j = 0: row jj itself is a copy) or jj-j (q-, v-grid, where only the second half of row
// jj is a copy), vector fields change sign.  The reference's three sweeps only move data, so every
// target (south rows, rows jj.., E/W strips of all those rows) is one composite gather; no source is
// itself a target.
__global__ void k_xctilr_arctic(const DevView *__restrict__ Vp, double *__restrict__ a, int nlev, int mhl, int nhl, int itype) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj;
  const int g = itype % 10;
  const double sgn = itype > 10 ? -1. : 1.;
  // targets: rows 1-nhl..0 and jj..jj+nhl over i = 1-mhl..ii+mhl, plus E/W strips of rows 1..jj-1
  const int wrow = ii + 2 * mhl, nrow = 2 * nhl + 1;
  const int nrowpts = nrow * wrow, nside = 2 * mhl * (jj - 1);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nrowpts + nside) return;
  int i, j;
  if (t < nrowpts) {
    const int r = t / wrow;
    i = t % wrow + 1 - mhl;
    j = r < nhl ? -r : jj + (r - nhl);            // 0,-1,..,1-nhl ; jj, jj+1, .., jj+nhl
  } else {
    t -= nrowpts;
    const int cidx = t % (2 * mhl);
    j = t / (2 * mhl) + 1;                        // 1..jj-1
    i = cidx < mhl ? -cidx : ii + (cidx - mhl) + 1;
  }
  const int iw = i < 1 ? i + ii : (i > ii ? i - ii : i);      // periodic in i
  bool land = false, flip = false;
  int is = iw, js = j;
  if (j < 1) land = true;
  else if (j >= jj) {
    const int d = j - jj;
    if (g == 1 || g == 3) {                                   // p-, u-grid
      is = g == 1 ? ii - (iw - 1) % ii : (ii - (iw - 1)) % ii + 1;
      js = jj - 1 - d;
      flip = true;
    } else if (d > 0 || iw > ii / 2) {                        // q-, v-grid
      is = g == 2 ? (ii - (iw - 1)) % ii + 1 : ii - (iw - 1) % ii;
      js = jj - d;
      flip = true;
    }
  }
  if (!land && !flip && i == iw) return;                      // first half of row jj on the q-/v-grid: not a target
  const size_t dst = IDX(V, i, j), src = IDX(V, is, js);
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    a[dst + o] = land ? V.P.vland : (flip ? sgn * a[src + o] : a[src + o]);
  }
}

// The same rule for a tile of a decomposed arctic domain (npx x npy tiles of one process, pointer
// transport): every target -- the tile's halo ring, and the seam row of the tiles in the last tile row --
// is mapped to the point of the GLOBAL domain it takes its value from (periodic wrap in i, fold across
// row jtdm), and read straight from the interior of the tile that owns that point.  With peer-mapped
// pointers the very same gather works across GPUs; no strip is packed or staged.
#define XCT_MAXTILES 64
#define XCT_MAXDIM 16
struct TileTab {
  double *p[XCT_MAXTILES];      // the array being updated, in every tile (index py*npx + px)
  int xoff[XCT_MAXDIM + 1];     // tile column q owns the global columns xoff[q]+1 .. xoff[q+1]   (bld/blom_dimensions:104-148:
  int yoff[XCT_MAXDIM + 1];     // tile rows likewise                                               tiles need not be equal)
};
// nreg decides what lies beyond the edges of the global domain: periodic wrap, land (vland), or -- nreg = 2 -- the fold
// across row jtdm (arctic = 1).  Tiles of unequal size: the owner of a global point is found in xoff / yoff, and its
// padded plane has its own row length and level stride.
__global__ void k_xctilr_tiles(const DevView *__restrict__ Vp, double *__restrict__ a, TileTab tab, int npx, int npy,
                               int px, int py, int nlev, int mhl, int nhl, int itype, int ew_per, int ns_per, int arctic) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj, itdm = tab.xoff[npx], jtdm = tab.yoff[npy];
  const int g = itype % 10;
  const double sgn = itype > 10 ? -1. : 1.;
  const int wrow = ii + 2 * mhl, nrow = jj + 2 * nhl;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= wrow * nrow) return;
  const int i = t % wrow + 1 - mhl, j = t / wrow + 1 - nhl;
  const int ig = tab.xoff[px] + i, jg = tab.yoff[py] + j;
  bool land = false, flip = false;
  int iw = ig;
  if (ig < 1 || ig > itdm) {
    if (ew_per) iw = ig < 1 ? ig + itdm : ig - itdm;
    else land = true;
  }
  int is = iw, js = jg;
  if (jg < 1) {
    if (ns_per) js = jg + jtdm;
    else land = true;
  } else if (arctic && jg >= jtdm) {
    const int d = jg - jtdm;
    if (g == 1 || g == 3) {                                              // p-, u-grid
      is = g == 1 ? itdm - (iw - 1) % itdm : (itdm - (iw - 1)) % itdm + 1;
      js = jtdm - 1 - d;
      flip = true;
    } else if (d > 0 || iw > itdm / 2) {                                 // q-, v-grid
      is = g == 2 ? (itdm - (iw - 1)) % itdm + 1 : itdm - (iw - 1) % itdm;
      js = jtdm - d;
      flip = true;
    }
  } else if (jg > jtdm) {
    if (ns_per) js = jg - jtdm;
    else land = true;
  }
  if (!land && !flip && i >= 1 && i <= ii && j >= 1 && j <= jj) return;  // interior point, not a target
  int qx = 0, qy = 0;
  if (!land) {
    while (qx + 1 < npx && is > tab.xoff[qx + 1]) qx++;
    while (qy + 1 < npy && js > tab.yoff[qy + 1]) qy++;
  }
  const double *src = land ? a : tab.p[qy * npx + qx];
  const int sni = tab.xoff[qx + 1] - tab.xoff[qx] + 2 * NBDY, snj = tab.yoff[qy + 1] - tab.yoff[qy] + 2 * NBDY;
  const size_t snp = (size_t)sni * snj;
  const size_t dst = IDX(V, i, j);
  const size_t so = land ? 0 : (size_t)(is - tab.xoff[qx] + NBDY - 1) + (size_t)sni * (js - tab.yoff[qy] + NBDY - 1);
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const double v = land ? V.P.vland : (flip ? sgn * src[so + (size_t)k * snp] : src[so + (size_t)k * snp]);
    a[dst + (size_t)k * V.nplane] = v;
  }
}

// ---- the arctic rule for tiles that cannot read each other's memory (one process per GPU) ---------------------
// Every tile of the top row packs its last `nrows` = nhl+2 interior rows (the only rows the fold reads: jtdm-1-nhl
// .. jtdm) into a strip [level][row][i]; the strips of the whole row are brought together (RCCL send/recv between
// the top-row ranks, comm_rccl.hip; a pointer table for tiles of one process), and each tile fills its fold targets
// -- rows jj.. over i = 1-mhl..ii+mhl, the E/W halo of those rows included -- from the strip of whichever tile owns
// the mirrored column.  Everything else (E/W, south) is the ordinary exchange, done before.  Same index rule as
// k_xctilr_arctic_tiles above, so the result is the single tile's.
struct StripTab {
  const double *p[XCT_MAXTILES];      // strip of top-row tile qx
};
struct PackSet { const double *a[4]; };
__global__ void k_arctic_pack(const DevView *__restrict__ Vp, PackSet P, double *__restrict__ strip0, int nlev, int nrows) {
  const DevView &V = *Vp;
  const double *__restrict__ a = P.a[blockIdx.z];
  double *__restrict__ strip = strip0 + (size_t)blockIdx.z * nrows * V.ii * nlev;
  const int ii = V.ii, jj = V.jj;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ii * nrows) return;
  const int i = t % ii + 1, r = t / ii, j = jj - nrows + 1 + r;
  const size_t src = IDX(V, i, j);
  for (int k = blockIdx.y; k < nlev; k += gridDim.y)
    strip[((size_t)k * nrows + r) * ii + (i - 1)] = j >= 1 ? a[src + (size_t)k * V.nplane] : 0.;
}
struct FillSet {                      // up to 4 stacks per launch (blockIdx.z): array, grid/field type, offset in a rank's block
  double *a[4];
  int itype[4];
  size_t off[4];
};
__global__ void k_arctic_fill(const DevView *__restrict__ Vp, FillSet F, StripTab tab, int npx, int px, int nlev, int mhl,
                              int nhl, int nrows) {
  const DevView &V = *Vp;
  double *__restrict__ a = F.a[blockIdx.z];
  const int itype = F.itype[blockIdx.z];
  const int ii = V.ii, jj = V.jj, itdm = npx * ii;
  const int g = itype % 10;
  const double sgn = itype > 10 ? -1. : 1.;
  const int wrow = ii + 2 * mhl;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= wrow * (nhl + 1)) return;
  const int i = t % wrow + 1 - mhl, d = t / wrow;                         // row jj + d
  const int ig = px * ii + i;
  const int iw = ig < 1 ? ig + itdm : (ig > itdm ? ig - itdm : ig);      // periodic in i
  int is, back;                                                          // source column, rows below the last one
  if (g == 1 || g == 3) {                                                // p-, u-grid
    is = g == 1 ? itdm - (iw - 1) % itdm : (itdm - (iw - 1)) % itdm + 1;
    back = 1 + d;
  } else if (d > 0 || iw > itdm / 2) {                                   // q-, v-grid
    is = g == 2 ? (itdm - (iw - 1)) % itdm + 1 : itdm - (iw - 1) % itdm;
    back = d;
  } else {
    return;                                                              // first half of the seam row: not a target
  }
  const int qx = (is - 1) / ii, r = nrows - 1 - back;
  const double *src = tab.p[qx] + F.off[blockIdx.z] + (size_t)r * ii + (is - qx * ii - 1);
  const size_t dst = IDX(V, i, jj + d);
  for (int k = blockIdx.y; k < nlev; k += gridDim.y)
    a[dst + (size_t)k * V.nplane] = sgn * src[(size_t)k * nrows * ii];
}

// strips of nf <= 4 stacks, one after the other, in one launch
void arctic_pack_launch(blomgpu_ctx *c, hipStream_t st, double *const *fields, int nf, double *strip, int nlev, int nrows) {
  const dim3 g((unsigned)((nrows * c->h.ii + 255) / 256), nlev > 64 ? 64 : nlev, nf);
  PackSet P;
  for (int f = 0; f < 4; f++) P.a[f] = fields[f < nf ? f : 0];
  hipLaunchKernelGGL(k_arctic_pack, g, dim3(256), 0, st, c->d, P, strip, nlev, nrows);
}

#include <pthread.h>
struct TileGroup {
  int npx, npy;
  std::vector<blomgpu_ctx *> tiles;      // index py*npx + px
  pthread_barrier_t bar;
};

int ctx_locate_ptr(const blomgpu_ctx *c, const double *p, size_t *offset) {
  for (int f = 0; f < NF_REAL; f++) {
    const double *b = c->h.f[f];
    const size_t n = (size_t)c->nlev_real[f] * c->h.nplane;
    if (p >= b && p < b + n) { *offset = (size_t)(p - b); return f; }
  }
  // the kk-level work space (stages that hand their result to the next one through it update its halos too): id NF_REAL
  const size_t nw = (size_t)c->h.nwk * c->h.kk * c->h.nplane;
  if (c->h.wk && p >= c->h.wk && p < c->h.wk + nw) { *offset = (size_t)(p - c->h.wk); return NF_REAL; }
  return -1;
}
static inline double *located_base(const blomgpu_ctx *c, int fid) { return fid == NF_REAL ? c->h.wk : c->h.f[fid]; }

// closed / periodic rule of the global domain in each direction (phy/mod_xc.F90:4378,4400)
static inline bool ew_periodic(int nreg) { return !(nreg == 0 || nreg == 4); }
static inline bool ns_periodic(int nreg) { return nreg > 2; }

// tiles of one process (TileGroup): every target is read from the interior of the tile that owns its source point
// fold: apply the arctic rule across row jtdm (otherwise the domain ends in land or wraps there, by nreg)
static int xctilr_group(blomgpu_ctx *c, double *a, int nlev, int mhl, int nhl, int itype, bool fold) {
  const DevView &h = c->h;
  const Tiling &T = c->tiling;
  TileGroup *G = T.group;
  if (T.npx * T.npy > XCT_MAXTILES || T.npx > XCT_MAXDIM || T.npy > XCT_MAXDIM) return ctx_fail(c, "xctilr: too many tiles for the in-process gather");
  if (fold && T.npx > 1 && T.npx % 2) return ctx_fail(c, "xctilr: the arctic patch needs an even number of tile columns (phy/mod_xc.F90:1600-1603)");
  if (nlev <= 0) return 0;
  size_t off = 0;
  const int fid = ctx_locate_ptr(c, a, &off);
  if (fid < 0) return ctx_fail(c, "xctilr: pointer does not belong to a registered field");
  // a level offset inside the owner's array: its planes have their own size
  const size_t lev = off / h.nplane;
  if (off % h.nplane) return ctx_fail(c, "xctilr: pointer is not the start of a plane");
  TileTab tab;
  for (int q = 0; q < XCT_MAXTILES; q++) {
    if (q >= T.npx * T.npy) { tab.p[q] = nullptr; continue; }
    const blomgpu_ctx *o = G->tiles[(size_t)q];
    tab.p[q] = located_base(o, fid) + lev * o->h.nplane;
  }
  for (int q = 0; q <= XCT_MAXDIM; q++) tab.xoff[q] = tab.yoff[q] = 0;
  for (int q = 0; q < T.npx; q++) { const DevView &o = G->tiles[(size_t)q]->h; tab.xoff[q] = o.i0; tab.xoff[q + 1] = o.i0 + o.ii; }
  for (int q = 0; q < T.npy; q++) { const DevView &o = G->tiles[(size_t)q * T.npx]->h; tab.yoff[q] = o.j0; tab.yoff[q + 1] = o.j0 + o.jj; }
  if (tab.xoff[T.npx] != h.itdm || tab.yoff[T.npy] != h.jtdm) return ctx_fail(c, "xctilr: the tiles of the group do not cover the global domain");
  HIPCHK(c, hipStreamSynchronize(c->stream));      // every tile's interior must be complete before anyone gathers
  pthread_barrier_wait(&G->bar);
  const int nt = (h.ii + 2 * mhl) * (h.jj + 2 * nhl);
  dim3 grid((nt + 255) / 256, nlev > 64 ? 64 : nlev);
  hipLaunchKernelGGL(k_xctilr_tiles, grid, dim3(256), 0, c->stream, c->d, a, tab, T.npx, T.npy, T.px, T.py, nlev, mhl, nhl, itype,
                     ew_periodic(h.nreg) ? 1 : 0, ns_periodic(h.nreg) ? 1 : 0, fold ? 1 : 0);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));      // ... and nobody may go on modifying its interior before all have read
  pthread_barrier_wait(&G->bar);
  return 0;
}

// the ordinary update: E/W and N/S neighbours (or the tile itself), land beyond closed boundaries
static int xctilr_plain(blomgpu_ctx *c, double *a, int nlev, int mhl, int nhl) {
  const DevView &h = c->h;
  const Tiling &T = c->tiling;
  const int ntarget = 2 * nhl * h.ii + 2 * mhl * (h.jj + 2 * nhl);
  if (ntarget == 0 || nlev <= 0) return 0;
  if (T.rccl) return rccl_xctilr(c, a, nlev, mhl, nhl);
  if (T.group) return xctilr_group(c, a, nlev, mhl, nhl, 1, false);
  if (h.itdm != h.ii || h.jtdm != h.jj)
    return ctx_fail(c, "xctilr: this context is one tile of a larger domain but no halo transport is attached");
  HaloSrc S;
  for (int dy = -1; dy <= 1; dy++)
    for (int dx = -1; dx <= 1; dx++) {
      const bool land = (dx != 0 && !ew_periodic(h.nreg)) || (dy != 0 && !ns_periodic(h.nreg));
      S.p[dy + 1][dx + 1] = land ? nullptr : a;                          // single tile: wraps onto itself
    }
  dim3 grid((ntarget + 255) / 256, nlev > 64 ? 64 : nlev);
  hipLaunchKernelGGL(k_xctilr_gather, grid, dim3(256), 0, c->stream, c->d, a, S, nlev, mhl, nhl);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int rccl_arctic_gather(blomgpu_ctx *c, double *const *fields, int nf, int nlev, int nrows, const double **strips, size_t *field_stride);
int rccl_xctilr_multi(blomgpu_ctx *c, double *const *fields, int nf, int nlev, int mhl, int nhl);

// arctic patch over tiles that exchange strips (RCCL ranks; tiles of one process with the option arctic_strips, which
// exists so that the pack/fill kernels can be tested for several tile columns on one GPU)
static int xctilr_arctic_strips(blomgpu_ctx *c, double *a, int nlev, int mhl, int nhl, int itype) {
  const DevView &h = c->h;
  const Tiling &T = c->tiling;
  if (T.npx > XCT_MAXTILES) return ctx_fail(c, "xctilr: too many tile columns for the arctic strips");
  if (int rc = xctilr_plain(c, a, nlev, mhl, nhl)) return rc;
  const bool top = T.py == T.npy - 1;
  const int nrows = nhl + 2;
  if (nrows > h.jj) return ctx_fail(c, "xctilr: tile has fewer rows than the arctic fold reads");
  StripTab tab;
  for (int q = 0; q < XCT_MAXTILES; q++) tab.p[q] = nullptr;
  const size_t need = (size_t)nrows * h.ii * nlev;
  if (T.rccl) {
    if (!top) return 0;
    size_t fs = 0;
    double *one[1] = {a};
    if (int rc = rccl_arctic_gather(c, one, 1, nlev, nrows, tab.p, &fs)) return rc;
  } else {
    TileGroup *G = T.group;
    if (top) {
      if (need > c->arc_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->arc_strip) (void)hipFree(c->arc_strip);
        HIPCHK(c, hipMalloc((void **)&c->arc_strip, need * sizeof(double)));
        c->arc_cap = need;
      }
      double *one[1] = {a};
      arctic_pack_launch(c, c->stream, one, 1, c->arc_strip, nlev, nrows);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    pthread_barrier_wait(&G->bar);                       // every strip is complete
    for (int q = 0; q < T.npx; q++) tab.p[q] = G->tiles[(size_t)(T.npy - 1) * T.npx + q]->arc_strip;
  }
  if (top) {
    const dim3 gf((unsigned)(((h.ii + 2 * mhl) * (nhl + 1) + 255) / 256), nlev > 64 ? 64 : nlev);
    FillSet F;
    for (int f = 0; f < 4; f++) { F.a[f] = a; F.itype[f] = itype; F.off[f] = 0; }
    hipLaunchKernelGGL(k_arctic_fill, gf, dim3(256), 0, c->stream, c->d, F, tab, T.npx, T.px, nlev, mhl, nhl, nrows);
    HIPCHK(c, hipGetLastError());
  }
  if (!T.rccl) {                                         // nobody repacks before everybody has read
    HIPCHK(c, hipStreamSynchronize(c->stream));
    pthread_barrier_wait(&T.group->bar);
  }
  return 0;
}

// up to 4 stacks of the same depth and halo widths over RCCL with the arctic patch: one E/W message per neighbour
// and one strip message per top-row rank for all of them (barotp's pb, ubflx, vbflx before every odd substep)
static int xctilr_arctic_rccl_multi(blomgpu_ctx *c, int nf, double *const *ptrs, int nlev, int mhl, int nhl, const int *itypes) {
  const DevView &h = c->h;
  const Tiling &T = c->tiling;
  if (T.npx > XCT_MAXTILES) return ctx_fail(c, "xctilr: too many tile columns for the arctic strips");
  const int ntarget = 2 * nhl * h.ii + 2 * mhl * (h.jj + 2 * nhl);
  if (ntarget > 0)
    if (int rc = rccl_xctilr_multi(c, ptrs, nf, nlev, mhl, nhl)) return rc;
  if (T.py != T.npy - 1) return 0;
  const int nrows = nhl + 2;
  if (nrows > h.jj) return ctx_fail(c, "xctilr: tile has fewer rows than the arctic fold reads");
  StripTab tab;
  for (int q = 0; q < XCT_MAXTILES; q++) tab.p[q] = nullptr;
  size_t fs = 0;
  if (int rc = rccl_arctic_gather(c, ptrs, nf, nlev, nrows, tab.p, &fs)) return rc;
  const dim3 gf((unsigned)(((h.ii + 2 * mhl) * (nhl + 1) + 255) / 256), nlev > 64 ? 64 : nlev, nf);
  FillSet F;
  for (int f = 0; f < 4; f++) { F.a[f] = ptrs[f < nf ? f : 0]; F.itype[f] = itypes[f < nf ? f : 0]; F.off[f] = (size_t)(f < nf ? f : 0) * fs; }
  // on the stream the exchange travels on (the second stream when it overlaps compute: stage_advect.hip)
  hipLaunchKernelGGL(k_arctic_fill, gf, dim3(256), 0, c->halo_stream ? c->halo_stream : c->stream, c->d, F, tab, T.npx, T.px, nlev, mhl, nhl, nrows);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_xctilr(blomgpu_ctx *c, double *base, int l1, int ld, int mh, int nh, int itype) {
  // itype only distinguishes grids/vectors across the arctic seam (phy/mod_xc.F90:4248-4250)
  const DevView &h = c->h;
  const int mhl = mh < 0 ? 0 : (mh > NBDY ? NBDY : mh);
  const int nhl = nh < 0 ? 0 : (nh > NBDY ? NBDY : nh);
  const int nlev = ld - l1 + 1;
  double *a = base + (size_t)(l1 - 1) * h.nplane;
  const Tiling &T = c->tiling;
  if (h.nreg == 2 && nlev > 0 && (T.rccl || (T.group && c->arctic_strips))) return xctilr_arctic_strips(c, a, nlev, mhl, nhl, itype);
  if (h.nreg == 2 && T.group) return xctilr_group(c, a, nlev, mhl, nhl, itype, true);   // decomposed arctic domain, all tiles in this process
  if (h.nreg == 2) {
    if (nlev <= 0) return 0;
    const int nt = (2 * nhl + 1) * (h.ii + 2 * mhl) + 2 * mhl * (h.jj - 1);
    dim3 grid((nt + 255) / 256, nlev > 64 ? 64 : nlev);
    hipLaunchKernelGGL(k_xctilr_arctic, grid, dim3(256), 0, c->stream, c->d, a, nlev, mhl, nhl, itype);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  return xctilr_plain(c, a, nlev, mhl, nhl);
}

// Several plane stacks with the same halo widths in one launch (single tile, no arctic patch): what the
// stages issue back to back (15 single planes before the barotropic loop, 6 stacks in halo_difest, ...).
// Other decompositions fall back to one update per stack.
#define XCT_MAXF 16
struct HaloMulti {
  double *p[XCT_MAXF];
  int nlev[XCT_MAXF];
};

__global__ void k_xctilr_multi(const DevView *__restrict__ Vp, HaloMulti M, int mhl, int nhl, int land_ew, int land_ns) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj;
  const int nns = 2 * nhl * ii;
  const int new_ = 2 * mhl * (jj + 2 * nhl);
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
  const int dx = i < 1 ? -1 : (i > ii ? 1 : 0), dy = j < 1 ? -1 : (j > jj ? 1 : 0);
  const bool land = (dx != 0 && land_ew) || (dy != 0 && land_ns);
  const size_t dst = IDX(V, i, j), src = IDX(V, i - dx * ii, j - dy * jj);
  double *a = M.p[blockIdx.z];
  const int nlev = M.nlev[blockIdx.z];
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    a[dst + o] = land ? V.P.vland : a[src + o];
  }
}

// The arctic rule for several plane stacks in one launch (single tile), each with its own grid/field
// type and halo widths: barotp's one-kernel-per-equation path updates pb (p scalar), ubflx (u vector) and
// vbflx (v vector, one more row) before every odd substep -- 63 times per baroclinic step on the
// tripolar grids.  Same composite gather as k_xctilr_arctic.
struct ArcticMulti {
  double *p[XCT_MAXF];
  int nlev[XCT_MAXF], itype[XCT_MAXF], mhl[XCT_MAXF], nhl[XCT_MAXF];
};
__global__ void k_xctilr_arctic_multi(const DevView *__restrict__ Vp, ArcticMulti M) {
  const DevView &V = *Vp;
  const int f = blockIdx.z;
  const int ii = V.ii, jj = V.jj, mhl = M.mhl[f], nhl = M.nhl[f], itype = M.itype[f];
  const int g = itype % 10;
  const double sgn = itype > 10 ? -1. : 1.;
  const int wrow = ii + 2 * mhl, nrow = 2 * nhl + 1;
  const int nrowpts = nrow * wrow, nside = 2 * mhl * (jj - 1);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nrowpts + nside) return;
  int i, j;
  if (t < nrowpts) {
    const int r = t / wrow;
    i = t % wrow + 1 - mhl;
    j = r < nhl ? -r : jj + (r - nhl);
  } else {
    t -= nrowpts;
    const int cidx = t % (2 * mhl);
    j = t / (2 * mhl) + 1;
    i = cidx < mhl ? -cidx : ii + (cidx - mhl) + 1;
  }
  const int iw = i < 1 ? i + ii : (i > ii ? i - ii : i);
  bool land = false, flip = false;
  int is = iw, js = j;
  if (j < 1) land = true;
  else if (j >= jj) {
    const int d = j - jj;
    if (g == 1 || g == 3) {
      is = g == 1 ? ii - (iw - 1) % ii : (ii - (iw - 1)) % ii + 1;
      js = jj - 1 - d;
      flip = true;
    } else if (d > 0 || iw > ii / 2) {
      is = g == 2 ? (ii - (iw - 1)) % ii + 1 : ii - (iw - 1) % ii;
      js = jj - d;
      flip = true;
    }
  }
  if (!land && !flip && i == iw) return;
  const size_t dst = IDX(V, i, j), src = IDX(V, is, js);
  double *a = M.p[f];
  const int nlev = M.nlev[f];
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    a[dst + o] = land ? V.P.vland : (flip ? sgn * a[src + o] : a[src + o]);
  }
}

// per-stack halo widths; single tile with the arctic patch only (the caller falls back otherwise)
int st_xctilr_arctic_multi(blomgpu_ctx *c, int nf, double *const *ptrs, const int *nlevs, const int *mhs, const int *nhs,
                           const int *itypes) {
  const DevView &h = c->h;
  if (h.nreg == 2 && c->tiling.rccl && nf >= 1 && nf <= 4) {
    bool same = nlevs[0] > 0;
    for (int f = 1; f < nf; f++) same = same && nlevs[f] == nlevs[0] && mhs[f] == mhs[0] && nhs[f] == nhs[0];
    if (same) {
      const int mhl = mhs[0] < 0 ? 0 : (mhs[0] > NBDY ? NBDY : mhs[0]), nhl = nhs[0] < 0 ? 0 : (nhs[0] > NBDY ? NBDY : nhs[0]);
      return xctilr_arctic_rccl_multi(c, nf, ptrs, nlevs[0], mhl, nhl, itypes);
    }
  }
  if (h.nreg != 2 || c->tiling.multi() || nf > XCT_MAXF) {
    for (int f = 0; f < nf; f++)
      if (int rc = st_xctilr(c, ptrs[f], 1, nlevs[f], mhs[f], nhs[f], itypes[f])) return rc;
    return 0;
  }
  if (nf <= 0) return 0;
  ArcticMulti M;
  int maxlev = 1, maxt = 0;
  for (int f = 0; f < XCT_MAXF; f++) {
    const bool on = f < nf;
    M.p[f] = on ? ptrs[f] : nullptr;
    M.nlev[f] = on ? nlevs[f] : 0;
    M.itype[f] = on ? itypes[f] : 1;
    M.mhl[f] = on ? (mhs[f] < 0 ? 0 : (mhs[f] > NBDY ? NBDY : mhs[f])) : 0;
    M.nhl[f] = on ? (nhs[f] < 0 ? 0 : (nhs[f] > NBDY ? NBDY : nhs[f])) : 0;
    if (on && nlevs[f] > maxlev) maxlev = nlevs[f];
    const int nt = (2 * M.nhl[f] + 1) * (h.ii + 2 * M.mhl[f]) + 2 * M.mhl[f] * (h.jj - 1);
    if (on && nt > maxt) maxt = nt;
  }
  dim3 grid((maxt + 255) / 256, maxlev > 64 ? 64 : maxlev, nf);
  hipLaunchKernelGGL(k_xctilr_arctic_multi, grid, dim3(256), 0, c->stream, c->d, M);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_xctilr_multi(blomgpu_ctx *c, int nf, double *const *ptrs, const int *nlevs, int mh, int nh, const int *itypes) {
  const DevView &h = c->h;
  if (h.nreg == 2 && !c->tiling.multi() && nf <= XCT_MAXF) {
    int mhs[XCT_MAXF], nhs[XCT_MAXF];
    for (int f = 0; f < nf; f++) { mhs[f] = mh; nhs[f] = nh; }
    return st_xctilr_arctic_multi(c, nf, ptrs, nlevs, mhs, nhs, itypes);
  }
  if (h.nreg == 2 && c->tiling.rccl) {                 // runs of stacks of equal depth share their messages
    for (int f = 0; f < nf;) {
      int g = 1;
      while (f + g < nf && g < 4 && nlevs[f + g] == nlevs[f]) g++;
      int mhs[4], nhs[4];
      for (int x = 0; x < g; x++) { mhs[x] = mh; nhs[x] = nh; }
      if (int rc = st_xctilr_arctic_multi(c, g, ptrs + f, nlevs + f, mhs, nhs, itypes + f)) return rc;
      f += g;
    }
    return 0;
  }
  if (c->tiling.rccl && h.nreg != 2) {                 // runs of stacks of equal depth travel in one message per neighbour
    const int mhl = mh < 0 ? 0 : (mh > NBDY ? NBDY : mh), nhl = nh < 0 ? 0 : (nh > NBDY ? NBDY : nh);
    if (2 * nhl * h.ii + 2 * mhl * (h.jj + 2 * nhl) == 0) return 0;
    for (int f = 0; f < nf;) {
      int g = 1;
      while (f + g < nf && g < 4 && nlevs[f + g] == nlevs[f]) g++;
      if (nlevs[f] > 0)
        if (int rc = rccl_xctilr_multi(c, ptrs + f, g, nlevs[f], mhl, nhl)) return rc;
      f += g;
    }
    return 0;
  }
  if (c->tiling.multi() || h.nreg == 2 || nf > XCT_MAXF) {
    for (int f = 0; f < nf; f++)
      if (int rc = st_xctilr(c, ptrs[f], 1, nlevs[f], mh, nh, itypes[f])) return rc;
    return 0;
  }
  const int mhl = mh < 0 ? 0 : (mh > NBDY ? NBDY : mh);
  const int nhl = nh < 0 ? 0 : (nh > NBDY ? NBDY : nh);
  const int ntarget = 2 * nhl * h.ii + 2 * mhl * (h.jj + 2 * nhl);
  if (ntarget == 0 || nf <= 0) return 0;
  if (h.itdm != h.ii || h.jtdm != h.jj)
    return ctx_fail(c, "xctilr: this context is one tile of a larger domain but no halo transport is attached");
  HaloMulti M;
  int maxlev = 1;
  for (int f = 0; f < XCT_MAXF; f++) { M.p[f] = f < nf ? ptrs[f] : nullptr; M.nlev[f] = f < nf ? nlevs[f] : 0; if (f < nf && nlevs[f] > maxlev) maxlev = nlevs[f]; }
  dim3 grid((ntarget + 255) / 256, maxlev > 64 ? 64 : maxlev, nf);
  hipLaunchKernelGGL(k_xctilr_multi, grid, dim3(256), 0, c->stream, c->d, M, mhl, nhl, ew_periodic(h.nreg) ? 0 : 1,
                     ns_periodic(h.nreg) ? 0 : 1);
  HIPCHK(c, hipGetLastError());
  return 0;
}

extern "C" {
// In-process tile group (tests of the decomposition on one device; one host thread per tile).
int blomgpu_group_create(int npx, int npy, TileGroup **out) {
  TileGroup *G = new TileGroup();
  G->npx = npx; G->npy = npy;
  G->tiles.assign((size_t)npx * npy, nullptr);
  pthread_barrier_init(&G->bar, nullptr, npx * npy);
  *out = G;
  return 0;
}
int blomgpu_group_attach(TileGroup *G, blomgpu_ctx *c, int px, int py) {
  if (px < 0 || px >= G->npx || py < 0 || py >= G->npy) return ctx_fail(c, "group_attach: tile index out of range");
  G->tiles[(size_t)py * G->npx + px] = c;
  c->tiling.npx = G->npx; c->tiling.npy = G->npy; c->tiling.px = px; c->tiling.py = py; c->tiling.group = G;
  return 0;
}
int blomgpu_group_destroy(TileGroup *G) {
  pthread_barrier_destroy(&G->bar);
  delete G;
  return 0;
}
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

__global__ void k_crc_strips(const DevView *__restrict__ Vp, const double *__restrict__ a, int nlev,
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

// ---- xccrc over several tiles (phy/mod_xc.F90:2195-2322) ---------------------------------------------------------
// The checksum is defined on the GLOBAL domain: rows are cut into strips of 2*nbdy+1 = 9 columns starting at global
// column 1, a strip's CRC runs over its wet points, a row's CRC chains its strips' CRCs, the result chains the rows'.
// A strip belongs to the tile that holds its centre column, which therefore finds all of it inside its halo of width
// nbdy after a halo update with vland = 0 (:2232-2240).  This entry does the halo update and returns the CRCs of the
// tile's own strips, [row 1..jj][strip], plus the global index of its first strip; the caller chains them over the
// tiles (blom_amd/tiles.py: chain_crc; between processes over torch.distributed) -- a diagnostic, off the hot path.
__global__ void k_crc_strips_tile(const DevView *__restrict__ Vp, const double *__restrict__ a, int nlev,
                                  const int *__restrict__ mask, int l0, int nstrip, unsigned *__restrict__ out) {
  const DevView &V = *Vp;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nstrip * V.jj) return;
  const int j = t / nstrip + 1, s = t % nstrip;
  const int g1 = 1 + (l0 + s) * (2 * NBDY + 1);                 // global columns g1..g2
  const int g2 = min(g1 + 2 * NBDY, V.itdm);
  unsigned crc8p = 0;
  for (int ig = g1; ig <= g2; ig++) {
    const int i = ig - V.i0;                                     // 1-nbdy .. ii+nbdy by the ownership rule
    if (mask[IDX(V, i, j)] != 1) continue;
    unsigned crc = ~crc8p;
    for (int k = 0; k < nlev; k++) {
      unsigned long long bits = __double_as_longlong(a[(size_t)k * V.nplane + IDX(V, i, j)]);
      for (int b = 0; b < 8; b++) crc = crc_byte(crc, (unsigned)(bits >> (8 * b)) & 255u);
    }
    crc8p = ~crc;
  }
  out[t] = crc8p;
}

int st_crc_strips(blomgpu_ctx *c, double *base, int nlev, int itype, unsigned *out, int cap, int *l0_out, int *ns_out) {
  DevView &h = c->h;
  const int g = itype % 10, W = 2 * NBDY + 1;
  const int *mask = g == 1 ? h.m[I_ip] : g == 2 ? h.m[I_iq] : g == 3 ? h.m[I_iu] : h.m[I_iv];
  // strips whose centre column (clipped to the domain) lies in i0+1 .. i0+ii
  const int ntot = (h.itdm + W - 1) / W;
  int l0 = -1, ns = 0;
  for (int l = 0; l < ntot; l++) {
    const int ctr = min(1 + l * W + NBDY, h.itdm);
    if (ctr > h.i0 && ctr <= h.i0 + h.ii) { if (l0 < 0) l0 = l; ns++; }
  }
  *l0_out = l0 < 0 ? 0 : l0;
  *ns_out = ns;
  if (ns * h.jj > cap) return ctx_fail(c, "crc_strips: output buffer too small");
  if (c->tiling.multi()) {                    // halo in i so that the strips are on chip (one tile: every strip is interior)
    // A plain E/W update (rows 1..jj, vland = 0): on a tripolar grid the fold is left out of it -- the serial xccrc
    // the golden checksums come from (phy/mod_xc.F90:4164-4205) reads the seam row as it stands, and so must the
    // strips that reach into a neighbour's columns (the reference's MPI form updates with the fold and restores the
    // seam row afterwards, :2233-2240, :2307-2311).
    const double vsave = h.P.vland;
    const int nreg_save = h.nreg;
    h.P.vland = 0.0;
    if (h.nreg == 2) h.nreg = 1;
    c->dirty = true; ctx_sync_view(c);
    int rc = st_xctilr(c, base, 1, nlev, NBDY, 0, itype % 10);
    h.P.vland = vsave; h.nreg = nreg_save; c->dirty = true; ctx_sync_view(c);
    if (rc) return rc;
  }
  if (ns == 0) return 0;                      // a narrow tile may own no strip; it still took part in the halo update
  unsigned *dout = nullptr;
  HIPCHK(c, hipMalloc((void **)&dout, sizeof(unsigned) * ns * h.jj));
  hipLaunchKernelGGL(k_crc_strips_tile, dim3((ns * h.jj + 63) / 64), dim3(64), 0, c->stream, c->d, base, nlev, mask, l0, ns, dout);
  HIPCHK(c, hipMemcpyAsync(out, dout, sizeof(unsigned) * ns * h.jj, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(dout);
  return 0;
}

// ---- xcsum (phy/mod_xc.F90:4116-4161): masked sum of a 2-D array that is reproducible bit for bit --------
// Every row is summed in strips of 2*nbdy+1 = 9 points, strip sums are added to the row sum in order, and
// the row sums are added serially: the order is part of the definition, so the rows go to the threads of ONE
// workgroup and thread 0 adds the row sums.  The mask of the global sums is ips (phy/mod_inigeo.F90:189-208):
// ip without the seam row of an arctic patch.  Single tile.
__global__ void k_xcsum(const DevView *__restrict__ Vp, const double *__restrict__ a, const int *__restrict__ mask, int skip_seam,
                        double *rowsum, double *out) {
  const DevView &V = *Vp;
  const int ii = V.ii, jj = V.jj;
  for (int j = 1 + (int)threadIdx.x; j <= jj; j += blockDim.x) {
    double sum8 = 0.;
    const bool dead = skip_seam && j >= jj;
    for (int i1 = 1; i1 <= ii; i1 += 2 * NBDY + 1) {
      double sum8p = 0.;
      const int ie = i1 + 2 * NBDY < ii ? i1 + 2 * NBDY : ii;
      for (int i = i1; i <= ie; i++) {
        const size_t x = IDX(V, i, j);
        if (!dead && mask[x] == 1) sum8p = sum8p + a[x];
      }
      sum8 = sum8 + sum8p;
    }
    rowsum[j - 1] = sum8;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = rowsum[0];
    for (int j = 2; j <= jj; j++) s = s + rowsum[j - 1];
    *out = s;
  }
}

// The same definition with the rows spread over the chip (thermf calls it twice per step: as one workgroup whose threads
// walk a row each it took 0.19 ms on the channel): a wavefront per row, lane s sums strip s, lane 0 adds the strip sums in
// order; then one workgroup adds the row sums in order.
__global__ __launch_bounds__(64) void k_xcsum_rows(const DevView *__restrict__ Vp, const double *__restrict__ a, const int *__restrict__ mask,
                                                  int skip_seam, double *__restrict__ rowsum) {
  const DevView &V = *Vp;
  __shared__ double strip[64];
  const int ii = V.ii, jj = V.jj, j = blockIdx.x + 1, W = 2 * NBDY + 1;
  const bool dead = skip_seam && j >= jj;
  const int nstrip = (ii + W - 1) / W;
  double sum8 = 0.;
  for (int s0 = 0; s0 < nstrip; s0 += 64) {
    const int sidx = s0 + (int)threadIdx.x;
    double sum8p = 0.;
    if (sidx < nstrip) {
      const int i1 = 1 + sidx * W, ie = i1 + 2 * NBDY < ii ? i1 + 2 * NBDY : ii;
      for (int i = i1; i <= ie; i++) {
        const size_t x = IDX(V, i, j);
        if (!dead && mask[x] == 1) sum8p = sum8p + a[x];
      }
    }
    strip[threadIdx.x] = sum8p;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = nstrip - s0 < 64 ? nstrip - s0 : 64;
      for (int q = 0; q < n; q++) sum8 = sum8 + strip[q];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) rowsum[j - 1] = sum8;
}

__global__ __launch_bounds__(256) void k_xcsum_total(const double *__restrict__ rowsum, int jj, double *__restrict__ out) {
  HIP_DYNAMIC_SHARED(double, rs)
  for (int j = threadIdx.x; j < jj; j += blockDim.x) rs[j] = rowsum[j];
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = rs[0];
    for (int j = 1; j < jj; j++) s = s + rs[j];
    *out = s;
  }
}

// Tiles of one process: every tile forms the sum of the WHOLE domain itself, reading each point from the tile that owns it --
// the strips, their order and the order of the rows are those of the global domain (phy/mod_xc.F90:1663-1708: a strip belongs to
// the tile that owns its centre, so the strip sums do not depend on the tiling), hence the single tile's bits on every tile.
struct XcsTab {
  const double *a[XCT_MAXTILES];
  const int *m[XCT_MAXTILES];
  int ni[XCT_MAXTILES];
  int xoff[XCT_MAXDIM + 1], yoff[XCT_MAXDIM + 1];
};
__global__ __launch_bounds__(64) void k_xcsum_rows_tiles(XcsTab tab, int npx, int npy, int itdm, int jtdm, int skip_seam, double *__restrict__ rowsum) {
  __shared__ double strip[64];
  const int j = blockIdx.x + 1, W = 2 * NBDY + 1;
  const bool dead = skip_seam && j >= jtdm;
  int py = 0;
  while (py + 1 < npy && j > tab.yoff[py + 1]) py++;
  const int nstrip = (itdm + W - 1) / W;
  double sum8 = 0.;
  for (int s0 = 0; s0 < nstrip; s0 += 64) {
    const int sidx = s0 + (int)threadIdx.x;
    double sum8p = 0.;
    if (sidx < nstrip) {
      const int i1 = 1 + sidx * W, ie = i1 + 2 * NBDY < itdm ? i1 + 2 * NBDY : itdm;
      int px = 0;
      for (int i = i1; i <= ie; i++) {
        while (px + 1 < npx && i > tab.xoff[px + 1]) px++;
        const int q = py * npx + px;
        const size_t x = (size_t)(j - tab.yoff[py] + NBDY - 1) * tab.ni[q] + (size_t)(i - tab.xoff[px] + NBDY - 1);
        if (!dead && tab.m[q][x] == 1) sum8p = sum8p + tab.a[q][x];
      }
    }
    strip[threadIdx.x] = sum8p;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = nstrip - s0 < 64 ? nstrip - s0 : 64;
      for (int q = 0; q < n; q++) sum8 = sum8 + strip[q];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) rowsum[j - 1] = sum8;
}

int rccl_xcsum_dev(blomgpu_ctx *c, const double *a, int itype, int slot);      // comm_rccl.hip

static int xcsum_group(blomgpu_ctx *c, const double *a, int itype, int slot) {
  const DevView &h = c->h;
  const Tiling &T = c->tiling;
  TileGroup *G = T.group;
  if (T.npx * T.npy > XCT_MAXTILES || T.npx > XCT_MAXDIM || T.npy > XCT_MAXDIM) return ctx_fail(c, "xcsum: too many tiles for the in-process gather");
  size_t off = 0;
  const int fid = ctx_locate_ptr(c, a, &off);
  if (fid < 0 || off % h.nplane) return ctx_fail(c, "xcsum: pointer is not a plane of a registered field");
  const size_t lev = off / h.nplane;
  const int g = itype % 10, mid = g == 1 ? I_ip : g == 2 ? I_iq : g == 3 ? I_iu : I_iv;
  XcsTab tab;
  for (int q = 0; q < XCT_MAXTILES; q++) {
    if (q >= T.npx * T.npy) { tab.a[q] = nullptr; tab.m[q] = nullptr; tab.ni[q] = 0; continue; }
    const blomgpu_ctx *o = G->tiles[(size_t)q];
    tab.a[q] = located_base(o, fid) + lev * o->h.nplane;
    tab.m[q] = o->h.m[mid];
    tab.ni[q] = o->h.ni;
  }
  for (int q = 0; q <= XCT_MAXDIM; q++) tab.xoff[q] = tab.yoff[q] = 0;
  for (int q = 0; q < T.npx; q++) { const DevView &o = G->tiles[(size_t)q]->h; tab.xoff[q] = o.i0; tab.xoff[q + 1] = o.i0 + o.ii; }
  for (int q = 0; q < T.npy; q++) { const DevView &o = G->tiles[(size_t)q * T.npx]->h; tab.yoff[q] = o.j0; tab.yoff[q + 1] = o.j0 + o.jj; }
  if (tab.xoff[T.npx] != h.itdm || tab.yoff[T.npy] != h.jtdm) return ctx_fail(c, "xcsum: the tiles of the group do not cover the global domain");
  HIPCHK(c, hipStreamSynchronize(c->stream));      // every tile's interior must be complete before anyone reads it
  pthread_barrier_wait(&G->bar);
  hipLaunchKernelGGL(k_xcsum_rows_tiles, dim3(h.jtdm), dim3(64), 0, c->stream, tab, T.npx, T.npy, h.itdm, h.jtdm, (g == 1 && h.nreg == 2) ? 1 : 0,
                     c->xcsum_buf + 1);
  hipLaunchKernelGGL(k_xcsum_total, dim3(1), dim3(256), sizeof(double) * (size_t)h.jtdm, c->stream, c->xcsum_buf + 1, h.jtdm, c->xcsum_dev + slot);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));      // ... and nobody may go on modifying its interior before all have read
  pthread_barrier_wait(&G->bar);
  return 0;
}

// the sum left on the device (slot of c->xcsum_dev): stages that only hand it to their next kernel stay capturable
int st_xcsum_dev(blomgpu_ctx *c, const double *a, int itype, int slot, double **sums_dev) {
  const DevView &h = c->h;
  if (!c->xcsum_buf) HIPCHK(c, hipMalloc((void **)&c->xcsum_buf, sizeof(double) * (size_t)(h.jtdm + 8)));
  if (!c->xcsum_dev) HIPCHK(c, hipMalloc((void **)&c->xcsum_dev, sizeof(double) * 8));
  *sums_dev = c->xcsum_dev;
  if (c->tiling.rccl) return rccl_xcsum_dev(c, a, itype, slot);
  if (c->tiling.group) return xcsum_group(c, a, itype, slot);
  if (h.itdm != h.ii || h.jtdm != h.jj) return ctx_fail(c, "xcsum: this context is one tile of a larger domain but no transport is attached");
  const int g = itype % 10;
  const int *mask = g == 1 ? h.m[I_ip] : g == 2 ? h.m[I_iq] : g == 3 ? h.m[I_iu] : h.m[I_iv];
  hipLaunchKernelGGL(k_xcsum_rows, dim3(h.jj), dim3(64), 0, c->stream, c->d, a, mask, (g == 1 && h.nreg == 2) ? 1 : 0, c->xcsum_buf + 1);
  hipLaunchKernelGGL(k_xcsum_total, dim3(1), dim3(256), sizeof(double) * (size_t)h.jj, c->stream, c->xcsum_buf + 1, h.jj, c->xcsum_dev + slot);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int st_xcsum(blomgpu_ctx *c, const double *a, int itype, double *sum) {
  const DevView &h = c->h;
  if (c->tiling.multi()) return ctx_fail(c, "xcsum with the result on the host: built for a single tile");
  if (!c->xcsum_buf) HIPCHK(c, hipMalloc((void **)&c->xcsum_buf, sizeof(double) * (size_t)(h.jtdm + 8)));
  const int g = itype % 10;
  const int *mask = g == 1 ? h.m[I_ip] : g == 2 ? h.m[I_iq] : g == 3 ? h.m[I_iu] : h.m[I_iv];
  hipLaunchKernelGGL(k_xcsum, dim3(1), dim3(256), 0, c->stream, c->d, a, mask, (g == 1 && h.nreg == 2) ? 1 : 0,
                     c->xcsum_buf + 1, c->xcsum_buf);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(sum, c->xcsum_buf, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
