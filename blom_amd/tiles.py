"""Uniform 2-D tile decomposition of a case (the scheme of bld/blom_dimensions:104-148 with equal
tile extents): tile (px,py) owns i0+1..i0+ii, j0+1..j0+jj.  A tile-local padded array is simply the
window [j0 : j0+jj+8, i0 : i0+ii+8] of the global padded array (local (i,j) <-> global (i0+i, j0+j))."""
import numpy as np


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
