"""Host-side plumbing of the one-process-per-GPU run: rank environment, tile layout along i,
distribution of the RCCL unique id, reductions of timings.  torch.distributed is used for
rendezvous and for scalar reductions only; the halo data path is RCCL point-to-point inside the
library (blom_amd/csrc/comm_rccl.hip).

Decomposition: the reference splits the global (itdm, jtdm) grid into tiles listed in
patch.input (bld/blom_dimensions:104-148); here tiles are equal and laid out along i, tile `rank`
owning global columns rank*idm+1 .. (rank+1)*idm.
"""
import os
from dataclasses import dataclass

import numpy as np


@dataclass
class RankEnv:
    rank: int
    world: int
    local: int


def rank_env(environ=None):
    e = os.environ if environ is None else environ
    return RankEnv(int(e.get("RANK", "0")), int(e.get("WORLD_SIZE", "1")), int(e.get("LOCAL_RANK", "0")))


def tile_layout(idm, rank, world):
    """(itdm, i0, px, npx) of the tile `rank` owns when `world` tiles of width idm lie along i."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return dict(itdm=idm * world, i0=rank * idm, px=rank, npx=world)


def neighbours(rank, world, periodic):
    """(west, east) ranks of a tile along i, -1 where the domain is closed
    (closed/periodic rule of phy/mod_xc.F90:4400-4416)."""
    west = rank - 1 if rank > 0 else (world - 1 if periodic else -1)
    east = rank + 1 if rank < world - 1 else (0 if periodic else -1)
    return west, east


def share_unique_id(make_id, env):
    """Rank 0 calls make_id() -> 128 bytes (ncclGetUniqueId); every rank returns the same bytes."""
    if env.world == 1:
        return make_id()
    import torch.distributed as dist
    box = [make_id() if env.rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


class FieldsFromRank0:
    """The fields of a whole-domain backend that exists on rank 0 only, served to every rank: field_names / field_info as the backend
    has them, get(name) = rank 0's array broadcast to all (a collective: every rank asks for the same names in the same order -- which
    blom_amd.tiles.scatter_to_tile and make_barotp_global do).  Tiles are initialised from it without the whole domain ever being
    built on the other ranks' GPUs.  device: "cuda:<n>" with the nccl backend (tensors travel GPU to GPU), "cpu" with gloo."""

    def __init__(self, whole, env, device="cpu"):
        import torch.distributed as dist
        self.whole, self.env, self.device = whole, env, device
        box = [None]
        if env.rank == 0:
            names = whole.field_names()
            box = [(names, {nm: whole.field_info(nm) for nm in names})]
        if env.world > 1:
            dist.broadcast_object_list(box, src=0)
        self._names, self._info = box[0]

    def field_names(self):
        return list(self._names)

    def field_info(self, name):
        return self._info[name]

    def has_field(self, name):
        return name in self._info

    def get(self, name):
        import numpy as np
        if self.env.world == 1:
            return self.whole.get(name)
        import torch
        import torch.distributed as dist
        # Rank 0 decides the element type and sends its torch name with the shape, so every rank builds the same tensor -- or all of
        # them fail here, BEFORE the collective (a type a rank could not map used to raise on the other ranks only, after rank 0 had
        # entered the broadcast, which then hung until its timeout).  The library's fields are float64 or int32; anything else a
        # backend hands out is converted to one of the two on rank 0.
        meta = [None]
        a = None
        if self.env.rank == 0:
            a = np.ascontiguousarray(self.whole.get(name))
            if a.dtype.kind in "iub" and a.dtype != np.int32:
                a = a.astype(np.int32)
            elif a.dtype.kind == "f" and a.dtype != np.float64:
                a = a.astype(np.float64)
            tname = {"float64": "float64", "int32": "int32"}.get(a.dtype.name)
            meta = [(a.shape, tname, None if tname else f"FieldsFromRank0: field {name!r} has dtype {a.dtype}")]
        dist.broadcast_object_list(meta, src=0)
        shape, tname, err = meta[0]
        if err:
            raise TypeError(err)
        # (the array comes back as a host numpy array on every rank: the windows of the tiles are cut on the host)
        t = torch.from_numpy(a) if self.env.rank == 0 else torch.empty(shape, dtype=getattr(torch, tname))
        if self.device != "cpu":
            t = t.to(self.device)
        dist.broadcast(t, src=0)
        return t.cpu().numpy()


def broadcast_object(obj, env):
    if env.world == 1:
        return obj
    import torch.distributed as dist
    box = [obj if env.rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def max_over_ranks(x, env, device="cpu"):
    if env.world == 1:
        return float(x)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_ints(v, env):
    if env.world == 1:
        return [int(v)]
    import torch.distributed as dist
    out = [None] * env.world
    dist.all_gather_object(out, int(v))
    return out


def all_gather_objects(obj, env):
    """every rank's `obj` (picklable), in rank order, on every rank"""
    if env.world == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * env.world
    dist.all_gather_object(out, obj)
    return out


def exchange_ew_host(a, idm, jdm, mhl, nhl, env, periodic):
    """Host (numpy + torch.distributed p2p) statement of the E/W phase of comm_rccl.hip, used by
    the gloo tests to check the message-order rule on real ranks: every rank sends west then
    east and receives east then west, so that with west == east (2 ranks, periodic) the peer's
    FIRST message lands in my east halo.  `a` is (nlev, jdm+8, idm+8), updated in place."""
    import torch
    import torch.distributed as dist
    west, east = neighbours(env.rank, env.world, periodic)
    rows = slice(4 - nhl, 4 + jdm + nhl)
    to_w = torch.from_numpy(np.ascontiguousarray(a[:, rows, 4:4 + mhl]))
    to_e = torch.from_numpy(np.ascontiguousarray(a[:, rows, 4 + idm - mhl:4 + idm]))
    from_e, from_w = torch.empty_like(to_w), torch.empty_like(to_e)
    ops = []
    if west >= 0:
        ops.append(dist.P2POp(dist.isend, to_w, west))
    if east >= 0:
        ops.append(dist.P2POp(dist.isend, to_e, east))
    if east >= 0:
        ops.append(dist.P2POp(dist.irecv, from_e, east))
    if west >= 0:
        ops.append(dist.P2POp(dist.irecv, from_w, west))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    vland = 0.0
    a[:, rows, 4 - mhl:4] = from_w.numpy() if west >= 0 else vland
    a[:, rows, 4 + idm:4 + idm + mhl] = from_e.numpy() if east >= 0 else vland
    return a


def tile_layout_2d(idm, jdm, rank, npx, npy):
    """window of tile `rank` = px + npx*py in a uniform npx x npy grid (bld/blom_dimensions:104-148)"""
    if not (0 <= rank < npx * npy):
        raise ValueError(f"rank {rank} outside a {npx}x{npy} tile grid")
    px, py = rank % npx, rank // npx
    return dict(itdm=idm * npx, jtdm=jdm * npy, i0=px * idm, j0=py * jdm, px=px, py=py, npx=npx, npy=npy)


def neighbours_2d(rank, npx, npy, periodic_i, periodic_j):
    """(west, east, south, north) ranks, -1 where the domain is closed -- the rule of
    rccl_xctilr_multi in blom_amd/csrc/comm_rccl.hip"""
    px, py = rank % npx, rank // npx
    row0 = rank - px
    west = rank - 1 if px > 0 else (row0 + npx - 1 if periodic_i else -1)
    east = rank + 1 if px < npx - 1 else (row0 if periodic_i else -1)
    south = rank - npx if py > 0 else (rank + npx * (npy - 1) if periodic_j else -1)
    north = rank + npx if py < npy - 1 else (rank - npx * (npy - 1) if periodic_j else -1)
    return west, east, south, north


def _sendrecv_pair(to_lo, to_hi, lo, hi):
    """send lo, send hi, receive hi, receive lo -- the matching order of comm_rccl.hip"""
    import torch
    import torch.distributed as dist
    from_hi, from_lo = torch.empty_like(to_lo), torch.empty_like(to_hi)
    ops = []
    if lo >= 0:
        ops.append(dist.P2POp(dist.isend, to_lo, lo))
    if hi >= 0:
        ops.append(dist.P2POp(dist.isend, to_hi, hi))
    if hi >= 0:
        ops.append(dist.P2POp(dist.irecv, from_hi, hi))
    if lo >= 0:
        ops.append(dist.P2POp(dist.irecv, from_lo, lo))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    return from_lo, from_hi


def exchange_2d_host(a, idm, jdm, mhl, nhl, rank, npx, npy, nreg):
    """Host (numpy + torch.distributed p2p) statement of both phases of rccl_xctilr_multi for an
    npx x npy tile grid: phase 1 rows of the columns 1..idm with south/north (tile-local for npy = 1),
    phase 2 the E/W strips over rows 1-nhl..jdm+nhl.  `a` is (nlev, jdm+8, idm+8), updated in place."""
    import torch
    periodic_i, periodic_j = nreg not in (0, 4), nreg > 2
    west, east, south, north = neighbours_2d(rank, npx, npy, periodic_i, periodic_j)
    vland = 0.0
    cols = slice(4, 4 + idm)
    if nhl > 0:
        if npy > 1:
            to_s = torch.from_numpy(np.ascontiguousarray(a[:, 4:4 + nhl, cols]))
            to_n = torch.from_numpy(np.ascontiguousarray(a[:, 4 + jdm - nhl:4 + jdm, cols]))
            from_s, from_n = _sendrecv_pair(to_s, to_n, south, north)
            a[:, 4 - nhl:4, cols] = from_s.numpy() if south >= 0 else vland
            a[:, 4 + jdm:4 + jdm + nhl, cols] = from_n.numpy() if north >= 0 else vland
        elif periodic_j:
            a[:, 4 - nhl:4, cols] = a[:, 4 + jdm - nhl:4 + jdm, cols]
            a[:, 4 + jdm:4 + jdm + nhl, cols] = a[:, 4:4 + nhl, cols]
        else:
            a[:, 4 - nhl:4, cols] = vland
            a[:, 4 + jdm:4 + jdm + nhl, cols] = vland
    if mhl > 0:
        rows = slice(4 - nhl, 4 + jdm + nhl)
        to_w = torch.from_numpy(np.ascontiguousarray(a[:, rows, 4:4 + mhl]))
        to_e = torch.from_numpy(np.ascontiguousarray(a[:, rows, 4 + idm - mhl:4 + idm]))
        if west == rank:            # one tile column, periodic: the wrap is tile-local (gloo has no self pair;
            from_w, from_e = to_e, to_w   # RCCL does send to itself there, tests/test_gpu_variants.py)
        else:
            from_w, from_e = _sendrecv_pair(to_w, to_e, west, east)
        a[:, rows, 4 - mhl:4] = from_w.numpy() if west >= 0 else vland
        a[:, rows, 4 + idm:4 + idm + mhl] = from_e.numpy() if east >= 0 else vland
    return a


def exchange_arctic_host(a, idm, jdm, mhl, nhl, rank, npx, npy, itype):
    """Host statement of the arctic update over ranks (comm_rccl.hip: rccl_arctic_gather; halo.hip: k_arctic_pack,
    k_arctic_fill): the ordinary exchange of a domain that is periodic in i and closed in j, then the ranks of the
    last tile row send each other their last nhl+2 interior rows and fill the fold targets -- rows jdm.. over the
    columns 1-mhl..idm+mhl -- from the strip of the tile that owns the mirrored column."""
    import torch
    import torch.distributed as dist
    exchange_2d_host(a, idm, jdm, mhl, nhl, rank, npx, npy, 2)
    px, py = rank % npx, rank // npx
    if py != npy - 1:
        return a
    nrows, row0 = nhl + 2, npx * (npy - 1)
    mine = torch.from_numpy(np.ascontiguousarray(a[:, 4 + jdm - nrows:4 + jdm, 4:4 + idm]))
    strips = [mine if q == px else torch.empty_like(mine) for q in range(npx)]
    ops = [dist.P2POp(dist.isend, mine, row0 + q) for q in range(npx) if q != px]
    ops += [dist.P2POp(dist.irecv, strips[q], row0 + q) for q in range(npx) if q != px]
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    strips = [s.numpy() for s in strips]
    itdm, g, sgn = npx * idm, itype % 10, (-1.0 if itype > 10 else 1.0)
    for d in range(nhl + 1):
        for i in range(1 - mhl, idm + mhl + 1):
            ig = px * idm + i
            iw = ig + itdm if ig < 1 else (ig - itdm if ig > itdm else ig)
            if g in (1, 3):
                src = itdm - (iw - 1) % itdm if g == 1 else (itdm - (iw - 1)) % itdm + 1
                back = 1 + d
            elif d > 0 or iw > itdm // 2:
                src = (itdm - (iw - 1)) % itdm + 1 if g == 2 else itdm - (iw - 1) % itdm
                back = d
            else:
                continue
            qx = (src - 1) // idm
            a[:, 3 + jdm + d, 3 + i] = sgn * strips[qx][:, nrows - 1 - back, src - qx * idm - 1]
    return a
