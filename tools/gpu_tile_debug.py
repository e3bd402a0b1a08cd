"""Stage-by-stage comparison of a tiled run (in-process transport) with the single tile (GPU box)."""
import sys, threading
import numpy as np
sys.path.insert(0, "tests")
from blom_amd.cases import make_case
from blom_amd.tiles import tile_extents, tile_window, scatter_state, gather_interior
from blom_amd.stepper import DYNCORE_STAGES
from blom_amd.hostinit import step_indices
from parity import STATE_FIELDS, GRID_FIELDS, INT_FIELDS, load_golden_init, put_fields
from blom_amd.gpu import BlomGpu, TileGroup

cfg, npx, npy = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ALL = STATE_FIELDS + GRID_FIELDS + INT_FIELDS
case = make_case(cfg)
masks, fields = load_golden_init(cfg)
ref = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, case.nreg, masks)
for nm, v in case.params.items():
    if not nm.endswith("0"):
        ref.set(nm, v)
put_fields(ref, fields)
ref.set("delt1", case.params["baclin"])
ii, jj = tile_extents(case, npx, npy)
grp = TileGroup(npx, npy)
tiles = {}
for py in range(npy):
    for px in range(npx):
        tm = {k: tile_window(masks[k], case, npx, npy, px, py) for k in masks}
        t = BlomGpu(ii, jj, case.kdm, case.ntr, case.nreg, tm, itdm=case.idm, jtdm=case.jdm, i0=px * ii, j0=py * jj)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                t.set(nm, v)
        t.set("delt1", case.params["baclin"])
        grp.attach(t, px, py)
        tiles[(px, py)] = t
scatter_state(ref, tiles, case, npx, npy, [f for f in ALL if f in fields])
names = [f for f in STATE_FIELDS if f in fields]
for nstep in range(1, 3):
    m, n, mm, nn, k1m, k1n = step_indices(nstep, case.kdm)
    for st in DYNCORE_STAGES:
        ref.stage(st, m, n, mm, nn, k1m, k1n)
        th = [threading.Thread(target=lambda t=t: (t.stage(st, m, n, mm, nn, k1m, k1n), t.sync())) for t in tiles.values()]
        [x.start() for x in th]; [x.join() for x in th]
        bad = []
        for nm in names:
            a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
            b = gather_interior(tiles, case, npx, npy, nm)
            if not np.array_equal(a, b, equal_nan=True):
                w = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
                bad.append((nm, len(w), tuple(w[0]), tuple(w[-1])))
        print(nstep, st, "OK" if not bad else bad[:8], flush=True)
        # halo ring of width W: tile padded array vs window of the single tile's padded array
        W = int(sys.argv[4]) if len(sys.argv) > 4 else 1
        hb = []
        for nm in names:
            a = ref.get(nm)
            for (px, py), t in tiles.items():
                b = t.get(nm)[:, 4 - W:4 + jj + W, 4 - W:4 + ii + W]
                aw = tile_window(a, case, npx, npy, px, py)[:, 4 - W:4 + jj + W, 4 - W:4 + ii + W]
                ne = ~((aw == b) | (np.isnan(aw) & np.isnan(b)))
                if ne.any():
                    w = np.argwhere(ne)
                    hb.append((nm, (px, py), len(w), sorted(set(int(v) for v in w[:, 0]))))
        if nstep == 2: print("   halo", hb, flush=True)
        if bad:
            sys.exit(1)
