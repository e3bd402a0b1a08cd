// xctilr over RCCL: one process per GPU, neighbour ncclSend/ncclRecv (xGMI is point-to-point, a
// halo exchange touches only the neighbouring GPUs).  Uniform npx x npy tile grid, rank = px + npx*py
// (the reference's mproc/nproc numbering, bld/blom_dimensions:104-148).  The reference's two phases
// (phy/mod_xc.F90:3036-3106 N-S, :3131-3178 E-W): phase 1 exchanges rows of the columns 1..ii with the
// south/north neighbours (tile-local when npy = 1: periodic wrap or vland), phase 2 exchanges the E/W
// strips over rows 1-nhl..jj+nhl, so the corners travel with it.  Messages: mhl*(jj+2*nhl)*nlev reals per
// neighbour, packed/unpacked by small kernels on the context's stream; RCCL runs on the same
// stream, so ordering with the stage kernels needs no events.
#include "blomgpu_internal.h"
#include <rccl/rccl.h>
#include <cstring>

struct RcclComm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  double *sbuf[2] = {nullptr, nullptr}, *rbuf[2] = {nullptr, nullptr};   // [0] west, [1] east
  size_t cap = 0;
  double *sbuf_ns[2] = {nullptr, nullptr}, *rbuf_ns[2] = {nullptr, nullptr};   // [0] south, [1] north
  size_t cap_ns = 0;
  double *arc_send = nullptr, *arc_gath = nullptr;   // arctic patch: my strip (self-send hook), the top row's strips
  size_t arc_cap = 0;
  int force_ns = 0;          // test hook: route a tile-local periodic N/S wrap through send/recv to itself
};

// up to MAXF plane stacks (same nlev, same halo widths) travel in one message per neighbour
#define MAXF 4
struct FieldSet {
  double *p[MAXF];
};

// One launch before the exchange does both of the reference's pre-send steps: phase 1 (N/S halo of
// columns 1..ii from the tile itself, periodic or vland -- blocks >= gpack) and the packing of the
// E/W strips (blocks < gpack): west buffer <- my west-most mhl interior columns (they become the west
// neighbour's east halo), east buffer <- my east-most mhl columns, rows 1-nhl..jj+nhl.  The strip rows
// outside 1..jj are the very values phase 1 writes, so the packer derives them itself instead of
// waiting for them.  Buffer layout [field][level][row][q].
__global__ void k_pack_ew_ns(const DevView *__restrict__ Vp, FieldSet F, double *__restrict__ west, double *__restrict__ east,
                             int nlev, int mhl, int nhl, int periodic, int gpack, int rows_present) {
  const DevView &V = *Vp;
  double *a = F.p[blockIdx.z];
  if ((int)blockIdx.x >= gpack) {
    const int t = (blockIdx.x - gpack) * blockDim.x + threadIdx.x;
    if (t >= 2 * nhl * V.ii) return;
    const int r = t / V.ii, i = t % V.ii + 1;
    const int j = r < nhl ? -r : V.jj + (r - nhl) + 1;
    const int js = j < 1 ? j + V.jj : j - V.jj;
    for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
      const size_t o = (size_t)k * V.nplane;
      a[IDX(V, i, j) + o] = periodic ? a[IDX(V, i, js) + o] : V.P.vland;
    }
    return;
  }
  const int nrow = V.jj + 2 * nhl, per = mhl * nrow;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per) return;
  const int q = t % mhl, r = t / mhl;          // q-th column of the strip, row index
  const int j = r + 1 - nhl;
  // rows_present: phase 1 was an exchange and its result is in the array; else derive the local rule
  const bool inside = rows_present || (j >= 1 && j <= V.jj);
  const int js = inside ? j : (j < 1 ? j + V.jj : j - V.jj);
  const bool land = !inside && !periodic;
  const size_t fo = (size_t)blockIdx.z * nlev * per;
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    west[fo + (size_t)k * per + t] = land ? V.P.vland : a[IDX(V, 1 + q, js) + o];                 // columns 1..mhl
    east[fo + (size_t)k * per + t] = land ? V.P.vland : a[IDX(V, V.ii - mhl + 1 + q, js) + o];    // ii-mhl+1..ii
  }
}

// unpack: from_west holds the west neighbour's east-most columns -> my columns 1-mhl..0;
//         from_east holds the east neighbour's west-most columns -> my columns ii+1..ii+mhl
__global__ void k_unpack_ew(const DevView *__restrict__ Vp, FieldSet F, const double *__restrict__ from_west,
                            const double *__restrict__ from_east, int nlev, int mhl, int nhl, int has_w, int has_e) {
  const DevView &V = *Vp;
  const int nrow = V.jj + 2 * nhl, per = mhl * nrow;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per) return;
  const int q = t % mhl, r = t / mhl;
  const int j = r + 1 - nhl;
  double *a = F.p[blockIdx.z];
  const size_t fo = (size_t)blockIdx.z * nlev * per;
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    a[IDX(V, 1 - mhl + q, j) + o] = has_w ? from_west[fo + (size_t)k * per + t] : V.P.vland;
    a[IDX(V, V.ii + 1 + q, j) + o] = has_e ? from_east[fo + (size_t)k * per + t] : V.P.vland;
  }
}

// phase 1 as an exchange (npy > 1): rows of the columns 1..ii.  south buffer <- my rows 1..nhl (they
// become the south neighbour's north halo), north buffer <- my rows jj-nhl+1..jj.  Layout [field][level][r][i].
__global__ void k_pack_ns(const DevView *__restrict__ Vp, FieldSet F, double *__restrict__ south, double *__restrict__ north,
                          int nlev, int nhl) {
  const DevView &V = *Vp;
  const int per = nhl * V.ii;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per) return;
  const int i = t % V.ii + 1, r = t / V.ii;
  const double *a = F.p[blockIdx.z];
  const size_t fo = (size_t)blockIdx.z * nlev * per;
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    south[fo + (size_t)k * per + t] = a[IDX(V, i, 1 + r) + o];
    north[fo + (size_t)k * per + t] = a[IDX(V, i, V.jj - nhl + 1 + r) + o];
  }
}
// from_south holds the south neighbour's north-most rows -> my rows 1-nhl..0; from_north the north
// neighbour's south-most rows -> my rows jj+1..jj+nhl; vland where the domain is closed
__global__ void k_unpack_ns(const DevView *__restrict__ Vp, FieldSet F, const double *__restrict__ from_south,
                            const double *__restrict__ from_north, int nlev, int nhl, int has_s, int has_n) {
  const DevView &V = *Vp;
  const int per = nhl * V.ii;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per) return;
  const int i = t % V.ii + 1, r = t / V.ii;
  double *a = F.p[blockIdx.z];
  const size_t fo = (size_t)blockIdx.z * nlev * per;
  for (int k = blockIdx.y; k < nlev; k += gridDim.y) {
    const size_t o = (size_t)k * V.nplane;
    a[IDX(V, i, 1 - nhl + r) + o] = has_s ? from_south[fo + (size_t)k * per + t] : V.P.vland;
    a[IDX(V, i, V.jj + 1 + r) + o] = has_n ? from_north[fo + (size_t)k * per + t] : V.P.vland;
  }
}

// `landed` != nullptr: leave the received E/W strips in the receive buffers (no unpack launch) and report
// them; the consumer reads its rim straight from there (barotp's substep-pair kernel).
int rccl_xctilr_multi_ex(blomgpu_ctx *c, double *const *fields, int nf, int nlev, int mhl, int nhl, RcclLanded *landed);
int rccl_xctilr_multi(blomgpu_ctx *c, double *const *fields, int nf, int nlev, int mhl, int nhl) {
  return rccl_xctilr_multi_ex(c, fields, nf, nlev, mhl, nhl, nullptr);
}
int rccl_xctilr_multi_ex(blomgpu_ctx *c, double *const *fields, int nf, int nlev, int mhl, int nhl, RcclLanded *landed) {
  const bool prepacked = landed && landed->prepacked;
  if (landed) { landed->from_west = landed->from_east = nullptr; landed->send_west = landed->send_east = nullptr; }
  const DevView &h = c->h;
  RcclComm *R = c->tiling.rccl;
  const Tiling &T = c->tiling;
  hipStream_t st = c->halo_stream ? c->halo_stream : c->stream;
  TimeScope tx(c, "exchange", st);         // pack, send/recv, unpack of this rank, on the stream they go to (bench.py: exchange_ms_per_rank)
  if (nf < 1 || nf > MAXF) return ctx_fail(c, "rccl_xctilr_multi: 1..4 fields per exchange");
  FieldSet F;
  for (int x = 0; x < MAXF; x++) F.p[x] = fields[x < nf ? x : 0];
  const int ly = nlev > 64 ? 64 : nlev;
  const int periodic_j = h.nreg > 2 ? 1 : 0;
  const bool per_i = !(h.nreg == 0 || h.nreg == 4);
  // ---- phase 1 as an exchange: several tile rows, or the single-rank test hook -----------------------
  const bool ns_exchange = nhl > 0 && (T.npy > 1 || (R->force_ns && periodic_j));
  if (ns_exchange) {
    const size_t per = (size_t)nhl * h.ii, need = per * nlev * nf;
    if (need > R->cap_ns) {
      HIPCHK(c, hipStreamSynchronize(st));
      for (int s = 0; s < 2; s++) {
        if (R->sbuf_ns[s]) (void)hipFree(R->sbuf_ns[s]);
        if (R->rbuf_ns[s]) (void)hipFree(R->rbuf_ns[s]);
        HIPCHK(c, hipMalloc((void **)&R->sbuf_ns[s], need * sizeof(double)));
        HIPCHK(c, hipMalloc((void **)&R->rbuf_ns[s], need * sizeof(double)));
      }
      R->cap_ns = need;
    }
    const int south = T.py > 0 ? R->rank - T.npx : (periodic_j ? R->rank + T.npx * (T.npy - 1) : -1);
    const int north = T.py < T.npy - 1 ? R->rank + T.npx : (periodic_j ? R->rank - T.npx * (T.npy - 1) : -1);
    dim3 g((unsigned)((per + 255) / 256), ly, nf);
    hipLaunchKernelGGL(k_pack_ns, g, dim3(256), 0, st, c->d, F, R->sbuf_ns[0], R->sbuf_ns[1], nlev, nhl);
    // same matching rule as E/W below: send south, send north, receive north, receive south
    ncclGroupStart();
    if (south >= 0) ncclSend(R->sbuf_ns[0], need, ncclDouble, south, R->comm, st);
    if (north >= 0) ncclSend(R->sbuf_ns[1], need, ncclDouble, north, R->comm, st);
    if (north >= 0) ncclRecv(R->rbuf_ns[1], need, ncclDouble, north, R->comm, st);
    if (south >= 0) ncclRecv(R->rbuf_ns[0], need, ncclDouble, south, R->comm, st);
    ncclResult_t rc = ncclGroupEnd();
    if (rc != ncclSuccess) return ctx_fail(c, std::string("RCCL halo exchange (N/S): ") + ncclGetErrorString(rc));
    hipLaunchKernelGGL(k_unpack_ns, g, dim3(256), 0, st, c->d, F, R->rbuf_ns[0], R->rbuf_ns[1], nlev, nhl,
                       south >= 0 ? 1 : 0, north >= 0 ? 1 : 0);
  }
  const unsigned gns = nhl > 0 && !ns_exchange ? (unsigned)((2 * nhl * h.ii + 255) / 256) : 0u;
  if (mhl <= 0) {
    if (gns)
      hipLaunchKernelGGL(k_pack_ew_ns, dim3(gns, ly, nf), dim3(256), 0, st, c->d, F, nullptr, nullptr, nlev, 0,
                         nhl, periodic_j, 0, 0);
  } else {
    const size_t per = (size_t)mhl * (h.jj + 2 * nhl), need = per * nlev * nf;
    bool regrown = false;          // buffers replaced in this call: strips a producer packed into the old ones are gone
    if (need > R->cap) {
      regrown = true;
      HIPCHK(c, hipStreamSynchronize(st));
      for (int s = 0; s < 2; s++) {
        if (R->sbuf[s]) (void)hipFree(R->sbuf[s]);
        if (R->rbuf[s]) (void)hipFree(R->rbuf[s]);
        HIPCHK(c, hipMalloc((void **)&R->sbuf[s], need * sizeof(double)));
        HIPCHK(c, hipMalloc((void **)&R->rbuf[s], need * sizeof(double)));
      }
      R->cap = need;
    }
    const int row0 = R->rank - T.px;        // first rank of my tile row
    const int west = T.px > 0 ? R->rank - 1 : (per_i ? row0 + T.npx - 1 : -1);
    const int east = T.px < T.npx - 1 ? R->rank + 1 : (per_i ? row0 : -1);
    const unsigned gpack = (unsigned)((per + 255) / 256);
    dim3 g(gpack, ly, nf);
    if (!prepacked || regrown)
      hipLaunchKernelGGL(k_pack_ew_ns, dim3(gpack + gns, ly, nf), dim3(256), 0, st, c->d, F, R->sbuf[0],
                         R->sbuf[1], nlev, mhl, nhl, periodic_j, (int)gpack, ns_exchange ? 1 : 0);
    // Message order matters when west == east (2 ranks periodic, or 1 rank sending to itself):
    // point-to-point operations between the same pair match in issue order, so every rank sends
    // west then east and receives east then west -- my east halo is the peer's FIRST send.
    ncclGroupStart();
    if (west >= 0) ncclSend(R->sbuf[0], need, ncclDouble, west, R->comm, st);
    if (east >= 0) ncclSend(R->sbuf[1], need, ncclDouble, east, R->comm, st);
    if (east >= 0) ncclRecv(R->rbuf[1], need, ncclDouble, east, R->comm, st);
    if (west >= 0) ncclRecv(R->rbuf[0], need, ncclDouble, west, R->comm, st);
    ncclResult_t rc = ncclGroupEnd();
    if (rc != ncclSuccess) return ctx_fail(c, std::string("RCCL halo exchange: ") + ncclGetErrorString(rc));
    if (landed) {
      landed->from_west = R->rbuf[0]; landed->from_east = R->rbuf[1];
      landed->has_w = west >= 0 ? 1 : 0; landed->has_e = east >= 0 ? 1 : 0;
      landed->per = (int)per; landed->mhl = mhl; landed->nhl = nhl; landed->nlev = nlev;
      landed->send_west = R->sbuf[0]; landed->send_east = R->sbuf[1];
    } else {
      hipLaunchKernelGGL(k_unpack_ew, g, dim3(256), 0, st, c->d, F, R->rbuf[0], R->rbuf[1], nlev, mhl, nhl,
                         west >= 0 ? 1 : 0, east >= 0 ? 1 : 0);
    }
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Arctic patch (nreg = 2): brings the strips of all tiles of the top row together on every rank of that row
// (halo.hip: k_arctic_pack / k_arctic_fill).  strips[q] <- device pointer of tile column q's strip.  The fold's
// mirror image of a tile's columns lies in the mirror tile and, for the u- and q-grid, one column beyond it
// (the reference's second "aia" partner, phy/mod_xc.F90:2640-2653), so instead of pairing tiles every top-row rank
// sends its strip ((nhl+2)*ii*nlev reals, a few 100 KB) to every other one: npx-1 messages per rank in one group.
void arctic_pack_launch(blomgpu_ctx *c, hipStream_t st, double *const *fields, int nf, double *strip, int nlev, int nrows);   // halo.hip
// Several plane stacks of the same depth travel in one message: rank q's block is [field][level][row][i].
int rccl_arctic_gather(blomgpu_ctx *c, double *const *fields, int nf, int nlev, int nrows, const double **strips, size_t *field_stride) {
  const DevView &h = c->h;
  RcclComm *R = c->tiling.rccl;
  const Tiling &T = c->tiling;
  hipStream_t st = c->halo_stream ? c->halo_stream : c->stream;
  TimeScope tx(c, "exchange", st);
  if (nf > 4) return ctx_fail(c, "rccl_arctic_gather: at most 4 stacks per exchange");
  const size_t per_field = (size_t)nrows * h.ii * nlev, need = per_field * nf;
  *field_stride = per_field;
  if (need > R->arc_cap) {
    HIPCHK(c, hipStreamSynchronize(st));
    if (R->arc_send) (void)hipFree(R->arc_send);
    if (R->arc_gath) (void)hipFree(R->arc_gath);
    HIPCHK(c, hipMalloc((void **)&R->arc_send, need * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&R->arc_gath, need * T.npx * sizeof(double)));
    R->arc_cap = need;
  }
  const bool self = T.npx == 1 && R->force_ns;           // test hook: my own strip travels through send/recv
  double *mine = self ? R->arc_send : R->arc_gath + (size_t)T.px * need;
  arctic_pack_launch(c, st, fields, nf, mine, nlev, nrows);
  if (T.npx > 1 || self) {
    const int row0 = T.npx * (T.npy - 1);
    ncclGroupStart();
    for (int q = 0; q < T.npx; q++)
      if (q != T.px || self) ncclSend(mine, need, ncclDouble, row0 + q, R->comm, st);
    for (int q = 0; q < T.npx; q++)
      if (q != T.px || self) ncclRecv(R->arc_gath + (size_t)q * need, need, ncclDouble, row0 + q, R->comm, st);
    ncclResult_t rc = ncclGroupEnd();
    if (rc != ncclSuccess) return ctx_fail(c, std::string("RCCL arctic exchange: ") + ncclGetErrorString(rc));
  }
  for (int q = 0; q < T.npx; q++) strips[q] = R->arc_gath + (size_t)q * need;
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- the barotropic solve replicated on every rank ---------------------------------------------------------------------
// barotp (phy/mod_barotp.F90:330-1003) subcycles 2.5 lstep times per baroclinic step with one halo exchange per odd+even
// pair; its substep kernel is bound by latency, not by the size of the domain (one CU per 32 x 16 tile, ~12 us per pair
// whatever the tile count), so cutting its 2-D domain into one piece per GPU buys nothing and puts ~60 exchanges per step
// on the critical path.  Instead every rank solves the WHOLE 2-D barotropic domain: the tiles bring the 2-D fields the
// solver reads together on every rank (one grouped send/recv per step), each rank runs the single tile's
// substep loop on a second context G that spans the global domain (kdm = 3: it holds 2-D fields), and takes its window of
// the results, halo included.  No exchange inside the loop.  Bit-identity with the decomposed solve is the reference's
// own guarantee (tiles compute redundantly into exchanged halos, so every tiling gives the single tile's bits).
static const int kBtFields[] = {
    F_uglue, F_vglue, F_umaxb, F_uminb, F_vmaxb, F_vminb, F_utotn, F_vtotn, F_pgfxm, F_pgfym, F_xixp, F_xixm, F_xiyp, F_xiym,
    F_pgfxm_o, F_pgfym_o, F_xixp_o, F_xixm_o, F_xiyp_o, F_xiym_o, F_pb, F_pbu, F_pbv, F_ub, F_vb, F_ubflxs, F_vbflxs,
    F_ubflxs_p, F_vbflxs_p, F_pb_p, F_pbu_p, F_pbv_p, F_ubcors_p, F_vbcors_p, F_ubflx, F_vbflx, F_pb_mn, F_ubflx_mn,
    F_vbflx_mn, F_pvtrop, F_pvtrop_o};
#define BT_MAXPLANES 96
struct BtPlanes {
  double *p[BT_MAXPLANES];
};
struct BtGlobal {
  blomgpu_ctx *G = nullptr;
  int nplanes = 0;
  std::vector<int> i0, j0, ii, jj;       // window of every rank
  std::vector<size_t> off;               // offset of rank q's block in the gather buffer (doubles)
  double *buf = nullptr;                 // all ranks' blocks, [rank][plane][j][i]
  int *geo_dev = nullptr;                // (i0, j0, ii, jj, offset/plane-stride) per rank for the unpack kernel
  int maxpts = 0;
};
// every listed plane of tile T, halo included -> its block
__global__ void k_btg_pack(const DevView *__restrict__ Vp, BtPlanes P, double *__restrict__ blk) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  blk[(size_t)blockIdx.y * V.nplane + t] = P.p[blockIdx.y][t];
}
// every rank's block -> the global planes of G (blockIdx.z = rank): a tile's interior points, and those of its halo
// points that lie in the halo of the GLOBAL domain (fields whose halos other stages filled -- pb_p, pbu_p.. -- are read
// there by the solver without another update, as on a single tile)
__global__ void k_btg_unpack(const DevView *__restrict__ Gp, BtPlanes P, const double *__restrict__ buf, const int *__restrict__ geo,
                             const size_t *__restrict__ offs) {
  const DevView &G = *Gp;
  const int q = blockIdx.z, i0 = geo[4 * q], j0 = geo[4 * q + 1], ii = geo[4 * q + 2], jj = geo[4 * q + 3];
  const int ni = ii + 2 * NBDY, n = ni * (jj + 2 * NBDY);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int i = t % ni - (NBDY - 1), j = t / ni - (NBDY - 1);
  const int gi = i0 + i, gj = j0 + j;
  // a halo point of the tile is taken where it lies beyond the edge of the global domain in that direction and inside the
  // tile's own range in the other (the corner pieces over a neighbour's rows / columns are that neighbour's to give)
  const bool x_ok = (i >= 1 && i <= ii) || gi < 1 || gi > G.ii;
  const bool y_ok = (j >= 1 && j <= jj) || gj < 1 || gj > G.jj;
  if (!x_ok || !y_ok) return;
  P.p[blockIdx.y][IDX(G, gi, gj)] = buf[offs[q] + (size_t)blockIdx.y * n + t];
}
// window of G (halo included) -> tile T
__global__ void k_btg_window(const DevView *__restrict__ Vp, BtPlanes PT, BtPlanes PG, int gni) {
  const DevView &V = *Vp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.nplane) return;
  const int x = t % V.ni, y = t / V.ni;
  PT.p[blockIdx.y][t] = PG.p[blockIdx.y][(size_t)(V.j0 + y) * gni + V.i0 + x];
}

blomgpu_ctx *bt_global_ctx(const blomgpu_ctx *c) { return c->bt_global ? c->bt_global->G : nullptr; }
int st_barotp_on(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n, bool with_bounds);     // stage_barotp.hip
int st_barotp_bounds(blomgpu_ctx *c, int m, int nn);

int rccl_barotp_replicated(blomgpu_ctx *c, int m, int n, int mm, int nn, int k1m, int k1n) {
  BtGlobal *B = c->bt_global;
  blomgpu_ctx *G = B->G;
  RcclComm *R = c->tiling.rccl;
  const DevView &h = c->h;
  hipStream_t st = c->stream;
  // the solver's scalar options follow the tile's (set through the tile's context only)
  if (memcmp(&G->h.P, &h.P, sizeof(Params)) != 0) { G->h.P = h.P; G->dirty = true; }
  G->defer_checks = c->defer_checks;
  ctx_sync_view(G);
  if (int rc = st_barotp_bounds(c, m, nn)) return rc;        // :177-224: the one part of barotp that reads 3-D fields
  BtPlanes PT, PG;
  int np_ = 0;
  for (int f : kBtFields)
    for (int l = 0; l < c->nlev_real[f]; l++) {
      PT.p[np_] = h.f[f] + (size_t)l * h.nplane;
      PG.p[np_] = G->h.f[f] + (size_t)l * G->h.nplane;
      np_++;
    }
  for (int x = np_; x < BT_MAXPLANES; x++) { PT.p[x] = PT.p[0]; PG.p[x] = PG.p[0]; }
  const int me = R->rank, nr = R->nranks;
  const int npts = h.nplane;
  TimeScope *tx = new TimeScope(c, "exchange", st);        // the gather of the replicated solve's planes: pack, send/recv, unpack
  hipLaunchKernelGGL(k_btg_pack, dim3((npts + 255) / 256, np_), dim3(256), 0, st, c->d, PT, B->buf + B->off[me]);
  if (nr > 1) {
    ncclGroupStart();
    for (int q = 0; q < nr; q++)
      if (q != me) ncclSend(B->buf + B->off[me], (size_t)npts * np_, ncclDouble, q, R->comm, st);
    for (int q = 0; q < nr; q++)
      if (q != me) ncclRecv(B->buf + B->off[q], (size_t)(B->ii[q] + 2 * NBDY) * (B->jj[q] + 2 * NBDY) * np_, ncclDouble, q, R->comm, st);
    ncclResult_t rc = ncclGroupEnd();
    if (rc != ncclSuccess) { delete tx; return ctx_fail(c, std::string("RCCL barotropic gather: ") + ncclGetErrorString(rc)); }
  }
  hipLaunchKernelGGL(k_btg_unpack, dim3((B->maxpts + 255) / 256, np_, nr), dim3(256), 0, st, G->d, PG, B->buf, B->geo_dev,
                     (const size_t *)(B->geo_dev + 4 * nr + (4 * nr) % 2));
  delete tx;
  if (int rc = st_barotp_on(G, m, n, mm, nn, k1m, k1n, false)) { c->err = G->err; return rc; }
  hipLaunchKernelGGL(k_btg_window, dim3((h.nplane + 255) / 256, np_), dim3(256), 0, st, c->d, PT, PG, G->h.ni);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// xcsum over RCCL ranks: the plane travels to every rank the way the barotropic solver's fields do (one grouped send/recv
// into the global context G of the replicated solve), and every rank forms the single tile's sum on G -- the reference's
// summation order is that of the global domain whatever the tiling (phy/mod_xc.F90:1663-1708, :2071-2192)
int st_xcsum_dev(blomgpu_ctx *c, const double *a, int itype, int slot, double **sums_dev);
int rccl_xcsum_dev(blomgpu_ctx *c, const double *a, int itype, int slot) {
  BtGlobal *B = c->bt_global;
  if (!B) return ctx_fail(c, "xcsum on RCCL tiles needs the global context of the replicated barotropic solve (blomgpu_rccl_attach_barotp_global)");
  blomgpu_ctx *G = B->G;
  RcclComm *R = c->tiling.rccl;
  const DevView &h = c->h;
  hipStream_t st = c->stream;
  size_t off = 0;
  const int fid = ctx_locate_ptr(c, a, &off);
  if (fid < 0 || fid >= NF_REAL || off % h.nplane) return ctx_fail(c, "xcsum: pointer is not a plane of a registered field");
  const size_t lev = off / h.nplane;
  if ((int)lev >= G->nlev_real[fid]) return ctx_fail(c, "xcsum: the global context does not hold this level");
  ctx_sync_view(G);
  BtPlanes PT, PG;
  for (int x = 0; x < BT_MAXPLANES; x++) { PT.p[x] = const_cast<double *>(a); PG.p[x] = G->h.f[fid] + lev * G->h.nplane; }
  const int me = R->rank, nr = R->nranks, npts = h.nplane;
  hipLaunchKernelGGL(k_btg_pack, dim3((npts + 255) / 256, 1), dim3(256), 0, st, c->d, PT, B->buf + B->off[me]);
  if (nr > 1) {
    ncclGroupStart();
    for (int q = 0; q < nr; q++)
      if (q != me) ncclSend(B->buf + B->off[me], (size_t)npts, ncclDouble, q, R->comm, st);
    for (int q = 0; q < nr; q++)
      if (q != me) ncclRecv(B->buf + B->off[q], (size_t)(B->ii[q] + 2 * NBDY) * (B->jj[q] + 2 * NBDY), ncclDouble, q, R->comm, st);
    ncclResult_t rc = ncclGroupEnd();
    if (rc != ncclSuccess) return ctx_fail(c, std::string("RCCL xcsum gather: ") + ncclGetErrorString(rc));
  }
  hipLaunchKernelGGL(k_btg_unpack, dim3((B->maxpts + 255) / 256, 1, nr), dim3(256), 0, st, G->d, PG, B->buf, B->geo_dev,
                     (const size_t *)(B->geo_dev + 4 * nr + (4 * nr) % 2));
  double *gs = nullptr;
  if (int rc = st_xcsum_dev(G, G->h.f[fid] + lev * G->h.nplane, itype, slot, &gs)) { c->err = G->err; return rc; }
  HIPCHK(c, hipMemcpyAsync(c->xcsum_dev + slot, gs + slot, sizeof(double), hipMemcpyDeviceToDevice, st));
  return 0;
}

int rccl_xctilr(blomgpu_ctx *c, double *a, int nlev, int mhl, int nhl) {
  return rccl_xctilr_multi(c, &a, 1, nlev, mhl, nhl);
}

extern "C" {
// 128-byte ncclUniqueId created on rank 0 and distributed by the launcher (torch.distributed in
// bench.py); every rank then joins the communicator.  Tiles: npx = nranks along i, npy = 1.
int blomgpu_rccl_unique_id(void *id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
  memcpy(id128, &id, sizeof(id) < 128 ? sizeof(id) : 128);
  return 0;
}
// Tiles: npx x npy, rank = px + npx*py, laid out as the reference's patch.input does it (bld/blom_dimensions:104-148;
// e.g. bld/tnx2v1/patch.input.8: four tile columns of 45, two tile rows of 97 and 96): every tile of a tile column
// has the same i-extent, every tile of a tile row the same j-extent, so E/W neighbours agree on the strip length and
// N/S neighbours on the row length without knowing each other's sizes.  The context must have been created with its
// window (i0, j0, ii, jj) of the (itdm, jtdm) domain; the tile columns must be equally wide on a tripolar grid (the
// fold pairs tile column q with npx-1-q).
int blomgpu_rccl_init_2d(blomgpu_ctx *c, const void *id128, int rank, int npx, int npy) {
  if (npx < 1 || npy < 1 || rank < 0 || rank >= npx * npy) return ctx_fail(c, "rccl_init: bad tile grid");
  const int px = rank % npx, py = rank / npx;
  const DevView &h = c->h;
  const bool first_i = px == 0, last_i = px == npx - 1, first_j = py == 0, last_j = py == npy - 1;
  if ((h.i0 == 0) != first_i || (h.i0 + h.ii == h.itdm) != last_i || (h.j0 == 0) != first_j || (h.j0 + h.jj == h.jtdm) != last_j ||
      h.i0 < 0 || h.j0 < 0 || h.i0 + h.ii > h.itdm || h.j0 + h.jj > h.jtdm)
    return ctx_fail(c, "rccl_init: the context's window (i0, j0, idm, jdm) does not fit tile (px,py) of the npx x npy grid");
  if (h.nreg == 2 && (h.i0 != px * h.ii || h.itdm != npx * h.ii))
    return ctx_fail(c, "rccl_init: a tripolar grid needs equally wide tile columns (i0 = px*idm, itdm = npx*idm)");
  if (h.ii < NBDY || h.jj < NBDY) return ctx_fail(c, "rccl_init: a tile must be at least nbdy points wide");
  HIPCHK(c, hipSetDevice(c->device));
  RcclComm *R = new RcclComm();
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id) < 128 ? sizeof(id) : 128);
  ncclResult_t rc = ncclCommInitRank(&R->comm, npx * npy, id, rank);
  if (rc != ncclSuccess) { delete R; return ctx_fail(c, std::string("ncclCommInitRank: ") + ncclGetErrorString(rc)); }
  R->rank = rank; R->nranks = npx * npy;
  c->tiling.rccl = R;
  if (!c->xstream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  c->tiling.npx = npx; c->tiling.npy = npy; c->tiling.px = px; c->tiling.py = py;
  return 0;
}
int blomgpu_rccl_init(blomgpu_ctx *c, const void *id128, int rank, int nranks) {      // tiles along i
  return blomgpu_rccl_init_2d(c, id128, rank, nranks, 1);
}
// test hook: with one tile row and a j-periodic domain, route the N/S wrap through send/recv (to the
// rank itself) instead of the local copy, so that the phase-1 exchange code runs on a single GPU
int blomgpu_rccl_force_ns_exchange(blomgpu_ctx *c, int on) {
  if (!c->tiling.rccl) return ctx_fail(c, "rccl_force_ns_exchange: no RCCL transport");
  c->tiling.rccl->force_ns = on;
  return 0;
}
// Replicated barotropic solve: G is a context of the GLOBAL domain (itdm x jtdm, kdm >= 3, same nreg, global masks and grid
// metrics uploaded by the caller) on the same device; isizes[npx], jsizes[npy] are the widths / heights of the tile columns
// and rows (bld/blom_dimensions:104-148).  From then on the tile's barotp gathers, solves on G and takes its window.
int blomgpu_rccl_attach_barotp_global(blomgpu_ctx *c, blomgpu_ctx *G, const int *isizes, const int *jsizes) {
  RcclComm *R = c->tiling.rccl;
  const Tiling &T = c->tiling;
  if (!R) return ctx_fail(c, "attach_barotp_global: no RCCL transport");
  if (!G || G->h.ii != c->h.itdm || G->h.jj != c->h.jtdm || G->h.nreg != c->h.nreg || G->tiling.multi() || G->device != c->device)
    return ctx_fail(c, "attach_barotp_global: the second context must span the global domain as a single tile on the same device");
  BtGlobal *B = new BtGlobal();
  B->G = G;
  for (int f : kBtFields) B->nplanes += c->nlev_real[f];
  if (B->nplanes > BT_MAXPLANES) { delete B; return ctx_fail(c, "attach_barotp_global: too many planes"); }
  const int nr = T.npx * T.npy;
  size_t off = 0;
  std::vector<int> geo;
  std::vector<size_t> offs;
  for (int q = 0; q < nr; q++) {
    const int qx = q % T.npx, qy = q / T.npx;
    int i0 = 0, j0 = 0;
    for (int x = 0; x < qx; x++) i0 += isizes[x];
    for (int y = 0; y < qy; y++) j0 += jsizes[y];
    B->i0.push_back(i0); B->j0.push_back(j0); B->ii.push_back(isizes[qx]); B->jj.push_back(jsizes[qy]);
    B->off.push_back(off);
    offs.push_back(off);
    geo.insert(geo.end(), {i0, j0, isizes[qx], jsizes[qy]});
    const int pts = (isizes[qx] + 2 * NBDY) * (jsizes[qy] + 2 * NBDY);       // a tile's planes travel with their halos
    off += (size_t)pts * B->nplanes;
    if (pts > B->maxpts) B->maxpts = pts;
  }
  if (B->i0[R->rank] != c->h.i0 || B->j0[R->rank] != c->h.j0 || B->ii[R->rank] != c->h.ii || B->jj[R->rank] != c->h.jj) {
    delete B;
    return ctx_fail(c, "attach_barotp_global: the tile sizes do not give this rank's window");
  }
  HIPCHK(c, hipMalloc((void **)&B->buf, off * sizeof(double)));
  const size_t ngeo = 4 * (size_t)nr + (4 * nr) % 2;          // keeps the size_t offsets behind it 8-byte aligned
  HIPCHK(c, hipMalloc((void **)&B->geo_dev, ngeo * sizeof(int) + nr * sizeof(size_t)));
  HIPCHK(c, hipMemcpy(B->geo_dev, geo.data(), geo.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(B->geo_dev + ngeo, offs.data(), nr * sizeof(size_t), hipMemcpyHostToDevice));
  // one stream for both contexts: the gather, the solve and the window copy are ordered by it
  HIPCHK(c, hipStreamSynchronize(G->stream));
  if (!G->stream_borrowed) (void)hipStreamDestroy(G->stream);
  G->stream = c->stream;
  G->stream_borrowed = true;
  c->bt_global = B;
  return 0;
}
int blomgpu_rccl_finalize(blomgpu_ctx *c) {
  RcclComm *R = c->tiling.rccl;
  if (c->bt_global) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->bt_global->buf);
    (void)hipFree(c->bt_global->geo_dev);
    delete c->bt_global;
    c->bt_global = nullptr;
  }
  if (!R) return 0;
  (void)hipStreamSynchronize(c->stream);
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  ncclCommDestroy(R->comm);
  for (int s = 0; s < 2; s++) {
    (void)hipFree(R->sbuf[s]); (void)hipFree(R->rbuf[s]);
    (void)hipFree(R->sbuf_ns[s]); (void)hipFree(R->rbuf_ns[s]);
  }
  if (R->arc_send) (void)hipFree(R->arc_send);
  if (R->arc_gath) (void)hipFree(R->arc_gath);
  delete R;
  c->tiling.rccl = nullptr;
  return 0;
}
}
