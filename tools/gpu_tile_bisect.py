"""Find which field's halo makes a stage of a tiled run differ from the single tile (GPU box).
usage: gpu_tile_bisect.py cfg npx npy nstep stage"""
import sys, threading
import numpy as np
sys.path.insert(0, "tests")
from blom_amd.cases import make_case
from blom_amd.tiles import tile_extents, tile_window, scatter_state, gather_interior
from blom_amd.stepper import DYNCORE_STAGES
from blom_amd.hostinit import step_indices
from parity import STATE_FIELDS, GRID_FIELDS, INT_FIELDS, load_golden_init, put_fields
from blom_amd.gpu import BlomGpu, TileGroup

cfg, npx, npy, NS, STG = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
ALL = STATE_FIELDS + GRID_FIELDS + INT_FIELDS
case = make_case(cfg)
masks, fields = load_golden_init(cfg)
def mk(ii, jj, m, **kw):
    t = BlomGpu(ii, jj, case.kdm, case.ntr, case.nreg, m, **kw)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            t.set(nm, v)
    t.set("delt1", case.params["baclin"])
    return t
ref = mk(case.idm, case.jdm, masks)
put_fields(ref, fields)
ii, jj = tile_extents(case, npx, npy)
grp = TileGroup(npx, npy)
tiles = {}
for py in range(npy):
    for px in range(npx):
        tm = {k: tile_window(masks[k], case, npx, npy, px, py) for k in masks}
        t = mk(ii, jj, tm, itdm=case.idm, jtdm=case.jdm, i0=px * ii, j0=py * jj)
        grp.attach(t, px, py)
        tiles[(px, py)] = t
scatter_state(ref, tiles, case, npx, npy, [f for f in ALL if f in fields])
names = [f for f in STATE_FIELDS if f in fields]

def run_tiles(st, idx):
    th = [threading.Thread(target=lambda t=t: (t.stage(st, *idx), t.sync())) for t in tiles.values()]
    [x.start() for x in th]; [x.join() for x in th]

W = int(sys.argv[6]) if len(sys.argv) > 6 else 0
WF = sys.argv[7].split(",") if len(sys.argv) > 7 else []
def mism():
    bad = []
    for nm in WF:
        a = ref.get(nm)
        for (px, py), t in tiles.items():
            b = t.get(nm)[:, 4 - W:4 + jj + W, 4 - W:4 + ii + W]
            aw = tile_window(a, case, npx, npy, px, py)[:, 4 - W:4 + jj + W, 4 - W:4 + ii + W]
            ne = ~((aw == b) | (np.isnan(aw) & np.isnan(b)))
            if ne.any():
                bad.append((nm, px, py, int(ne.sum())))
    if WF:
        return bad
    for nm in names:
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior(tiles, case, npx, npy, nm)
        if not np.array_equal(a, b, equal_nan=True):
            bad.append(nm)
    return bad

done = False
for nstep in range(1, NS + 1):
    idx = step_indices(nstep, case.kdm)
    for st in DYNCORE_STAGES:
        if nstep == NS and st == STG:
            done = True
            break
        ref.stage(st, *idx)
        run_tiles(st, idx)
    if done:
        break
snap_ref = {nm: ref.get(nm).copy() for nm in names}
snap_t = {k: {nm: t.get(nm).copy() for nm in names} for k, t in tiles.items()}
def restore():
    for nm in names:
        ref.put(nm, snap_ref[nm])
        for k, t in tiles.items():
            t.put(nm, snap_t[k][nm])
ref.stage(STG, *idx); run_tiles(STG, idx)
print("baseline mismatch:", mism(), flush=True)
for F in names:
    restore()
    for (px, py), t in tiles.items():
        t.put(F, tile_window(snap_ref[F], case, npx, npy, px, py))
    ref.stage(STG, *idx); run_tiles(STG, idx)
    b = mism()
    if not b:
        print("FIXED by syncing halo of", F, flush=True)
