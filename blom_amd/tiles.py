"""2-D tile decomposition of a case in the reference's patch.input scheme (bld/blom_dimensions:104-148): tile columns of
widths iipe(px), tile rows of heights jjpe(py); tile (px,py) owns the global points i0+1..i0+ii, j0+1..j0+jj.  A
tile-local padded array is simply the window [j0 : j0+jj+8, i0 : i0+ii+8] of the global padded array (local (i,j) <->
global (i0+i, j0+j)), halo included -- which is why a state scattered this way needs no halo update before the first step.
`chain_crc` completes the decomposition-independent checksum xccrc (phy/mod_xc.F90:2195-2322) from the strip CRCs of
the tiles."""
import re
import zlib
from dataclasses import dataclass

import numpy as np

NBDY = 4


@dataclass
class TileLayout:
    isizes: tuple      # widths of the tile columns
    jsizes: tuple      # heights of the tile rows

    @property
    def npx(self):
        return len(self.isizes)

    @property
    def npy(self):
        return len(self.jsizes)

    @property
    def itdm(self):
        return int(sum(self.isizes))

    @property
    def jtdm(self):
        return int(sum(self.jsizes))

    def tile(self, px, py):
        """(i0, j0, ii, jj) of tile (px, py)"""
        return int(sum(self.isizes[:px])), int(sum(self.jsizes[:py])), int(self.isizes[px]), int(self.jsizes[py])

    def rank_tile(self, rank):
        """rank = px + npx*py (the reference's mproc/nproc numbering)"""
        return rank % self.npx, rank // self.npx

    def window(self, a, px, py):
        i0, j0, ii, jj = self.tile(px, py)
        return np.ascontiguousarray(a[..., j0:j0 + jj + 2 * NBDY, i0:i0 + ii + 2 * NBDY])

    @staticmethod
    def split(n, parts):
        """sizes of `parts` tiles covering n points the way bld/blom_dimensions cuts them: the first tiles one longer"""
        q, r = divmod(n, parts)
        return tuple(q + (1 if k < r else 0) for k in range(parts))

    @classmethod
    def regular(cls, idm, jdm, npx, npy):
        return cls(cls.split(idm, npx), cls.split(jdm, npy))

    @classmethod
    def from_patch_input(cls, path):
        """tile sizes of a reference patch.input file (iipe of the first tile row, jjpe of the first tile column);
        only tensor-product layouts (every tile row cut alike) are carried"""
        txt = open(path).read()
        rows = re.findall(r"iipe\(\s*\d+\)\s*=\s*([\d\s]+)", txt)
        cols = re.findall(r"jjpe\(\s*\d+\)\s*=\s*([\d\s]+)", txt)
        ii = [tuple(int(x) for x in r.split()) for r in rows]
        jj = [tuple(int(x) for x in c.split()) for c in cols]
        if any(r != ii[0] for r in ii) or any(c != jj[0] for c in jj):
            raise ValueError(f"{path}: tile rows are cut differently (land-tile elimination); not a tensor-product layout")
        return cls(ii[0], jj[0])


def tile_extents(case, npx, npy):
    assert case.idm % npx == 0 and case.jdm % npy == 0, "uniform tiles only"
    return case.idm // npx, case.jdm // npy


def tile_window(a, case, npx, npy, px, py):
    ii, jj = tile_extents(case, npx, npy)
    i0, j0 = px * ii, py * jj
    return np.ascontiguousarray(a[..., j0:j0 + jj + 8, i0:i0 + ii + 8])


def scatter_state(src, tiles, case, npx, npy, fields):
    """Copy `fields` of the single-tile backend `src` into the tile backends (dict (px,py)->backend)."""
    for nm in fields:
        try:
            a = src.get(nm)
        except KeyError:
            continue
        for (px, py), t in tiles.items():
            if t.has_field(nm):
                t.put(nm, tile_window(a, case, npx, npy, px, py))


def scatter_to_tile(src, tile, layout, px, py, fields=None, skip=("mpack",)):
    """Copy every field (default: all the library registers) of the whole-domain backend `src` into tile (px,py)."""
    for nm in (fields or src.field_names()):
        if nm in skip or nm.startswith("wkp"):
            continue
        tile.put(nm, layout.window(src.get(nm), px, py))


def gather_interior(tiles, case, npx, npy, name):
    """Assemble the global interior (nlev, jdm, idm) of field `name` from the tiles."""
    ii, jj = tile_extents(case, npx, npy)
    out = None
    for (px, py), t in tiles.items():
        a = t.get(name)
        if out is None:
            out = np.zeros((a.shape[0], case.jdm, case.idm), a.dtype)
        out[:, py * jj:(py + 1) * jj, px * ii:(px + 1) * ii] = a[:, 4:4 + jj, 4:4 + ii]
    return out


def gather_interior_layout(tiles, layout, name):
    out = None
    for (px, py), t in tiles.items():
        i0, j0, ii, jj = layout.tile(px, py)
        a = t.get(name)
        if out is None:
            out = np.zeros((a.shape[0], layout.jtdm, layout.itdm), a.dtype)
        out[:, j0:j0 + jj, i0:i0 + ii] = a[:, 4:4 + jj, 4:4 + ii]
    return out


def chain_crc(parts, layout):
    """xccrc of the whole domain from the tiles' strip CRCs.  parts[(px,py)] = (l0, strips[jj, ns]) as returned by
    BlomGpu.crc_strips: a row's CRC chains its strips in global order, the result chains the rows in global order
    (phy/mod_xc.F90:2262-2300)."""
    rows = []
    for py in range(layout.npy):
        jj = layout.jsizes[py]
        row_parts = sorted((parts[(px, py)] for px in range(layout.npx)), key=lambda t: t[0])
        for j in range(jj):
            crc8 = 0
            for _, strips in row_parts:
                for v in strips[j]:
                    crc8 = zlib.crc32(np.array([v], dtype="<u4").tobytes(), crc8)
            rows.append(crc8)
    return zlib.crc32(np.array(rows, dtype="<u4").tobytes()) & 0xFFFFFFFF


def make_barotp_global(src, case, masks, device=0):
    """The second context of an RCCL tile that solves the whole 2-D barotropic domain (blomgpu_rccl_attach_barotp_global):
    global dimensions, kdm = 3 (it holds 2-D fields only), no tracers, the global masks, and every 2-D field of the
    whole-domain backend `src` (grid metrics, the barotropic state at start)."""
    from .gpu import BlomGpu
    g = BlomGpu(case.idm, case.jdm, 3, 0, case.nreg, masks, device=device)
    for nm in src.field_names():
        if nm == "mpack" or nm.startswith("wkp") or nm in masks:
            continue
        nlev, isint = src.field_info(nm)
        if nlev <= 3 and not isint and g.has_field(nm) and g.field_info(nm)[0] == nlev:
            g.put(nm, src.get(nm))
    return g
