"""Decomposition independence on the device: the same case integrated as 1 tile and as several tiles
(in-process transport, one thread per tile, all on one GPU) must give bit-identical interiors --
the property the reference guarantees by computing redundantly into exchanged halos.
Also: the RCCL transport with a single rank (periodic direction wraps onto the rank itself) must
reproduce the plain single-tile halo update.  tri_s: the arctic patch decomposed (the seam row and its halo
are gathered from the mirror tiles through the tile pointer table)."""
import threading

import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd.tiles import tile_extents, tile_window, scatter_state, gather_interior
from parity import STATE_FIELDS, GRID_FIELDS, INT_FIELDS, load_golden_init, put_fields

pytestmark = pytest.mark.gpu
from blom_amd.hostinit import step_indices as hostinit_step_indices
ALL = STATE_FIELDS + GRID_FIELDS + INT_FIELDS
CHECK = ["u", "v", "dp", "temp", "saln", "sigma", "pb", "ub", "vb", "ubflxs_p", "pb_p", "trc", "uflx", "vflx",
         "pgfx", "pgfy", "dpu", "dpv", "pbu", "pbv", "ubflx", "vbflx", "pb_mn", "ubcors_p"]


def _single(cfg, nsteps):
    import os
    from blom_amd.gpu import BlomGpu
    from blom_amd import hostinit
    case = make_case(cfg)
    here = os.path.dirname(os.path.abspath(__file__))
    if os.path.exists(os.path.join(here, "golden", f"{cfg}_init.npz")):
        masks, fields = load_golden_init(cfg)
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, case.nreg, masks)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        put_fields(gpu, fields)
    else:                                   # no committed fixture: initialise on the single tile itself
        nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
        masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
        hostinit.init_state(gpu, case)
        fields = {nm: gpu.get(nm) for nm in ALL if gpu.has_field(nm)}
    gpu.set("delt1", case.params["baclin"])
    return case, masks, fields, gpu


@pytest.mark.parametrize("cfg,npx,npy", [("chan_s", 2, 1), ("chan_s", 1, 2), ("chan_s", 2, 2), ("box_s", 2, 2),
                                         ("box_s", 3, 1), ("tri_s", 2, 1), ("tri_s", 2, 2), ("tri_s", 4, 2),
                                         ("chan_s_tke", 2, 2), ("tri_s_tke", 2, 2),
                                         # arctic fold through packed strips -- the kernels of the RCCL transport's
                                         # arctic exchange, here with the strips handed over by pointer
                                         ("tri_s+strips", 2, 1), ("tri_s+strips", 2, 2), ("tri_s+strips", 4, 2),
                                         ("tri_s+strips", 3, 1), ("tri_s_tke+strips", 4, 1)])
def test_tiles_match_single_tile(cfg, npx, npy):
    from blom_amd.gpu import BlomGpu, TileGroup
    nsteps = 4
    strips = cfg.endswith("+strips")
    cfg = cfg.split("+")[0]
    case, masks, fields, ref = _single(cfg, nsteps)
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
            t.set("arctic_strips", 1 if strips else 0)
            grp.attach(t, px, py)
            tiles[(px, py)] = t
    scatter_state(ref, tiles, case, npx, npy, [f for f in ALL if f in fields])
    assert ref.step(0, nsteps) == nsteps
    errs = []

    def run(t):
        try:
            t.step(0, nsteps)
            t.sync()
        except Exception as e:          # a failing tile would leave the others at a barrier
            errs.append(e)
    th = [threading.Thread(target=run, args=(t,)) for t in tiles.values()]
    [x.start() for x in th]
    [x.join(timeout=300) for x in th]
    assert not errs, errs
    bad = []
    for nm in CHECK:
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior(tiles, case, npx, npy, nm)
        if not np.array_equal(a, b):
            bad.append((nm, int((a != b).sum()), float(np.nanmax(np.abs(a - b)))))
    for t in tiles.values():
        t.close()
    ref.close()
    grp.destroy()
    assert not bad, bad


@pytest.mark.parametrize("cfg,npx,npy", [("chan_s", 2, 2), ("box_s", 2, 2), ("tri_s", 2, 1), ("tri_s", 2, 2), ("tri_s_tke", 4, 2)])
def test_cppm_on_tiles_matches_single_tile(cfg, npx, npy):
    """advect with advmth = 'cppm' on a decomposed domain: every tile builds its coefficient tables (init_cppm,
    phy/mod_cppm.F90:2504-2722, halo updates of tags and edge coefficients included) and steps; on a tripolar grid the tiles
    that hold the seam swap tags and edge coefficients across it (:2650-2722, :1531-1541, :1686-1704) -- the rule is written
    in global indices.  Interiors as on the single tile."""
    from blom_amd.gpu import BlomGpu, TileGroup
    from blom_amd import hostinit
    nsteps = 4
    case = make_case(cfg, advmth="cppm")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    ref = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(ref, case)
    fields = [nm for nm in ALL if ref.has_field(nm)]
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
    scatter_state(ref, tiles, case, npx, npy, fields)
    assert ref.step(0, nsteps) == nsteps
    errs = []
    kk = case.kdm

    def run(t):
        try:
            t.stage("init_cppm", 2, 1, kk, 0, kk + 1, 1)
            t.step(0, nsteps)
            t.sync()
        except Exception as e:          # a failing tile would leave the others at a barrier
            errs.append(e)
    th = [threading.Thread(target=run, args=(t,), daemon=True) for t in tiles.values()]
    [x.start() for x in th]
    [x.join(timeout=300) for x in th]
    assert not errs, errs
    bad = []
    for nm in CHECK:
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior(tiles, case, npx, npy, nm)
        if not np.array_equal(a, b):
            bad.append((nm, int((a != b).sum()), float(np.nanmax(np.abs(a - b)))))
    for t in tiles.values():
        t.close()
    ref.close()
    grp.destroy()
    assert not bad, bad


@pytest.mark.parametrize("cfg,npx,npy,vcoord,method,neutral", [
    ("chan_s", 2, 2, "cntiso_hybrid", "nudge", 0), ("box_s", 2, 2, "cntiso_hybrid", "direct", 0), ("tri_s", 2, 2, "cntiso_hybrid", "nudge", 0),
    ("tri_s", 4, 2, "plevel", "direct", 0), ("chan_s", 2, 1, "plevel", "nudge", 0),
    # ltedtp = 'neutral': the regridding covers two rings beyond the tile (neutral diffusion + lateral smoothing), the searches
    # of the faces on the tile's edge read the neighbour's columns from the halo
    ("chan_s", 2, 2, "cntiso_hybrid", "nudge", 1), ("box_s", 2, 2, "cntiso_hybrid", "direct", 1), ("tri_s", 2, 2, "cntiso_hybrid", "nudge", 1),
    ("tri_s", 4, 2, "plevel", "direct", 1)])
def test_ale_regrid_remap_on_tiles_matches_single_tile(cfg, npx, npy, vcoord, method, neutral):
    """ale_regrid_remap (phy/mod_ale_regrid_remap.F90:1486) on a decomposed domain: every tile has its own engine structures; the
    lateral smoothing of regrid_method = 'nudge' regrids one ring of columns beyond the tile from halo data and the stage's own
    halo updates go through the tile transport.  After a few steps of the isopycnic sequence: interiors as on the single tile."""
    from blom_amd.gpu import BlomGpu, TileGroup
    nsteps = 3
    case, masks, fields, ref = _single(cfg, nsteps)
    kk = case.kdm
    ii, jj = tile_extents(case, npx, npy)
    grp = TileGroup(npx, npy)
    tiles = {}
    for py in range(npy):
        for px in range(npx):
            tm = {k: tile_window(masks[k], case, npx, npy, px, py) for k in masks}
            t = BlomGpu(ii, jj, kk, case.ntr, case.nreg, tm, itdm=case.idm, jtdm=case.jdm, i0=px * ii, j0=py * jj)
            for nm, v in case.params.items():
                if not nm.endswith("0"):
                    t.set(nm, v)
            t.set("delt1", case.params["baclin"])
            grp.attach(t, px, py)
            tiles[(px, py)] = t
    scatter_state(ref, tiles, case, npx, npy, [f for f in ALL if f in fields])
    assert ref.step(0, nsteps) == nsteps
    pbot = float(np.max(ref.get("p")[kk][masks["ip"] > 0]))
    plevel = 0.4 * pbot * (np.arange(kk) / kk) ** 1.3
    six = hostinit_step_indices(nsteps, kk)

    dpml = 9806.0 * (20.0 + 60.0 * np.linspace(0.0, 1.0, case.jdm + 8)[:, None] * np.ones((1, case.idm + 8)))[None]
    if neutral:
        ref.put("dpml", dpml)
        scatter_state(ref, tiles, case, npx, npy, ["dpml"])

    def ale(g):
        g.set("vcoord_type", vcoord)
        g.set("ale_regrid_method", method)
        g.set_vector("plevel", plevel)
        if neutral:
            g.set("ltedtp_opt", 2)
            g.set("ndiff_surface_align", 1)
        g.stage("ale_regrid_remap", *six)
    ale(ref)
    errs = []

    def run(t):
        try:
            t.step(0, nsteps)
            ale(t)
            t.sync()
        except Exception as e:          # a failing tile would leave the others at a barrier
            errs.append(e)
    th = [threading.Thread(target=run, args=(t,), daemon=True) for t in tiles.values()]
    [x.start() for x in th]
    [x.join(timeout=300) for x in th]
    assert not errs, errs
    bad = []
    for nm in ["dp", "temp", "saln", "sigma", "trc", "u", "v", "dpu", "dpv", "p", "pu", "pv"] + \
            (["utflld", "usflld", "vtflld", "vsflld", "utflx", "usflx", "vtflx", "vsflx", "nslpx", "nslpy"] if neutral else []):
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior(tiles, case, npx, npy, nm)
        if not np.array_equal(a, b):
            bad.append((nm, int((a != b).sum()), float(np.nanmax(np.abs(a - b)))))
    for t in tiles.values():
        t.close()
    ref.close()
    grp.destroy()
    assert not bad, bad


HYBRID_INPUTS = ["kvisc_m", "kdiff_t", "kdiff_s", "t_ns_nonloc", "s_nb_nonloc", "t_rs_nonloc", "s_rs_nonloc", "mu_nonloc", "mv_nonloc",
                 "sswflx", "surflx", "surrlx", "brnflx", "salflx", "salrlx", "swfc1", "swfc2", "swal1", "swal2", "OBLdepth", "trflx"]


def hybrid_inputs(case, seed=11):
    """what the routines the hybrid step leaves out would produce (diffusivities, non-local fractions, surface fluxes, absorption
    bands, boundary layer depth): smooth synthetic fields"""
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    rng = np.random.default_rng(seed)
    z = np.arange(kk + 1)[:, None, None] / kk
    f = {}
    for nm in ("kvisc_m", "kdiff_t", "kdiff_s"):
        f[nm] = 1e-5 + 10.0 ** rng.uniform(-3.5, -2.0, (1, nj, ni)) * np.exp(-((z - 0.1) / 0.15) ** 2)
    for nm in ("t_ns_nonloc", "s_nb_nonloc", "t_rs_nonloc", "s_rs_nonloc", "mu_nonloc", "mv_nonloc"):
        a = np.clip(1.0 - z / rng.uniform(0.1, 0.6, (1, nj, ni)), 0.0, 1.0) ** 2
        a[0] = 1.0
        f[nm] = a
    f["sswflx"] = -rng.uniform(0.0, 150.0, (1, nj, ni))
    f["surflx"] = f["sswflx"] + rng.uniform(-100.0, 100.0, (1, nj, ni))
    f["surrlx"] = rng.uniform(-10.0, 10.0, (1, nj, ni))
    f["brnflx"] = -rng.uniform(0.0, 1e-4, (1, nj, ni))
    f["salflx"] = f["brnflx"] + rng.uniform(-2e-3, 2e-3, (1, nj, ni))
    f["salrlx"] = rng.uniform(-5e-4, 5e-4, (1, nj, ni))
    f["swfc1"] = rng.uniform(0.4, 0.7, (1, nj, ni))
    f["swfc2"] = 1.0 - f["swfc1"]
    f["swal1"] = rng.uniform(0.5, 1.5, (1, nj, ni))
    f["swal2"] = rng.uniform(10.0, 20.0, (1, nj, ni))
    f["OBLdepth"] = 10.0 ** rng.uniform(0.8, 2.2, (1, nj, ni))
    if case.ntr:
        f["trflx"] = rng.uniform(-1e-6, 1e-6, (case.ntr, nj, ni))
    return f


@pytest.mark.parametrize("cfg,npx,npy,vcoord,method,advmth,neutral", [
    ("chan_s", 2, 2, "cntiso_hybrid", "nudge", "remap", 0), ("box_s", 2, 2, "cntiso_hybrid", "direct", "cppm", 0),
    ("tri_s", 2, 2, "cntiso_hybrid", "nudge", "remap", 0), ("tri_s", 4, 2, "plevel", "direct", "cppm", 0), ("chan_s", 1, 2, "plevel", "nudge", "remap", 0),
    # with ltedtp = 'neutral' (neutral diffusion inside ale_regrid_remap, its slopes through cmnfld_nnslope_ale into eddtra_ale)
    ("chan_s", 2, 2, "cntiso_hybrid", "nudge", "remap", 1), ("tri_s", 2, 2, "cntiso_hybrid", "nudge", "remap", 1),
    ("box_s", 2, 2, "cntiso_hybrid", "direct", "cppm", 1)])
def test_hybrid_step_on_tiles_matches_single_tile(cfg, npx, npy, vcoord, method, advmth, neutral):
    """The step of the hybrid vertical coordinate (ale_regrid_remap, cmnfld2's hybrid branches, eddtra_ale, advect,
    .., ale_forcing, ale_vdifft/m, .., cmnfld1) on a decomposed domain: every halo update of the new stages -- the ring of the
    lateral smoothing, the bounded mixed layer depth and the mixed layer density of eddtra_ale, the viscosity of ale_vdiffm, the
    slopes and buoyancy frequencies of cmnfld -- goes through the tile transport.  Interiors after a few steps as on the single tile."""
    from blom_amd.gpu import BlomGpu, TileGroup
    nsteps = 3
    case = make_case(cfg, advmth=advmth)
    _, masks, fields, ref = _single(cfg, nsteps)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            ref.set(nm, v)
    ref.set("delt1", case.params["baclin"])
    kk = case.kdm
    f = hybrid_inputs(case)
    # the tiles get windows of the same padded arrays, halo points included: whatever a stage reads there without updating it
    # first is the same number on both sides
    for nm, a in f.items():
        ref.put(nm, a)
    pbot = float(np.max(ref.get("p")[kk][masks["ip"] > 0]))
    plevel = 0.3 * pbot * (np.arange(kk) / kk) ** 1.3
    ii, jj = tile_extents(case, npx, npy)
    grp = TileGroup(npx, npy)
    tiles = {}
    for py in range(npy):
        for px in range(npx):
            tm = {k: tile_window(masks[k], case, npx, npy, px, py) for k in masks}
            t = BlomGpu(ii, jj, kk, case.ntr, case.nreg, tm, itdm=case.idm, jtdm=case.jdm, i0=px * ii, j0=py * jj)
            for nm, v in case.params.items():
                if not nm.endswith("0"):
                    t.set(nm, v)
            t.set("delt1", case.params["baclin"])
            grp.attach(t, px, py)
            tiles[(px, py)] = t
    scatter_state(ref, tiles, case, npx, npy, [nm for nm in ALL if nm in fields] + [nm for nm in HYBRID_INPUTS if nm in f])
    six0 = hostinit_step_indices(0, kk)

    def run_hybrid(g):
        g.set("vcoord_type", vcoord)
        g.set("ale_regrid_method", method)
        g.set("mlrmth", "fox08")
        g.set("swamxd", 200.0)
        g.set("brine_mlbase_frac", 0.4)
        g.set_vector("plevel", plevel)
        g.set("ltedtp_opt", 2 if neutral else 1)
        g.set("ndiff_surface_align", 1)
        if advmth == "cppm":
            g.stage("init_cppm", 2, 1, kk, 0, kk + 1, 1)
        g.stage("cmnfld1", *six0)
        assert g.step(0, nsteps) == nsteps
    run_hybrid(ref)
    errs = []

    def run(t):
        try:
            run_hybrid(t)
            t.sync()
        except Exception as e:          # a failing tile would leave the others at a barrier
            errs.append(e)
    th = [threading.Thread(target=run, args=(t,), daemon=True) for t in tiles.values()]
    [x.start() for x in th]
    [x.join(timeout=300) for x in th]
    assert not errs, errs
    bad = []
    for nm in CHECK + ["umfltd", "vmfltd", "umflsm", "vmflsm", "hml_tf", "mld", "bfsqi", "nslpx", "nslpy", "salt_corr", "buoyfl"] + \
            (["nnslpx", "nnslpy", "utflld", "vtflld"] if neutral else []):
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior(tiles, case, npx, npy, nm)
        wet = np.isfinite(a) & (np.abs(a) < 1e30)
        if not np.array_equal(a[wet], b[wet]):
            bad.append((nm, int((a[wet] != b[wet]).sum()), float(np.nanmax(np.abs(a[wet] - b[wet])))))
    for t in tiles.values():
        t.close()
    ref.close()
    grp.destroy()
    assert not bad, bad


@pytest.mark.parametrize("cfg,isizes,jsizes", [("box_s", (12, 12), (11, 9)), ("chan_s", (7, 7, 6), (13, 11)),
                                               ("tri_s", (6, 6, 6, 6), (11, 9)), ("tri_s_tke", (12, 12), (9, 11))])
def test_unequal_tiles_match_single_tile(cfg, isizes, jsizes):
    """Tiles of unequal size in the reference's patch.input scheme (bld/blom_dimensions:104-148: tile columns / rows need
    not be equally wide; bld/tnx2v1/patch.input.8 has rows of 97 and 96) with the in-process transport: the gather finds the
    owner of every halo point in the tables of tile offsets and reads it with the owner's plane geometry."""
    from blom_amd.gpu import BlomGpu, TileGroup
    from blom_amd.tiles import TileLayout, scatter_to_tile, gather_interior_layout
    nsteps = 4
    case, masks, fields, ref = _single(cfg, nsteps)
    lay = TileLayout(tuple(isizes), tuple(jsizes))
    assert lay.itdm == case.idm and lay.jtdm == case.jdm
    grp = TileGroup(lay.npx, lay.npy)
    tiles = {}
    for py in range(lay.npy):
        for px in range(lay.npx):
            i0, j0, ii, jj = lay.tile(px, py)
            t = BlomGpu(ii, jj, case.kdm, case.ntr, case.nreg, {k: lay.window(masks[k], px, py) for k in masks},
                        itdm=case.idm, jtdm=case.jdm, i0=i0, j0=j0)
            for nm, v in case.params.items():
                if not nm.endswith("0"):
                    t.set(nm, v)
            scatter_to_tile(ref, t, lay, px, py)
            t.set("delt1", case.params["baclin"])
            grp.attach(t, px, py)
            tiles[(px, py)] = t
    assert ref.step(0, nsteps) == nsteps
    errs = []

    def run(t):
        try:
            t.step(0, nsteps)
            t.sync()
        except Exception as e:
            errs.append(e)
    th = [threading.Thread(target=run, args=(t,), daemon=True) for t in tiles.values()]
    [x.start() for x in th]
    [x.join(timeout=300) for x in th]
    assert not errs, errs
    assert not any(x.is_alive() for x in th), "a tile did not finish"
    bad = []
    for nm in CHECK:
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior_layout(tiles, lay, nm)
        if not np.array_equal(a, b):
            bad.append((nm, int((a != b).sum())))
    for t in tiles.values():
        t.close()
    ref.close()
    grp.destroy()
    assert not bad, bad


def test_rccl_transport_single_rank_equals_single_tile():
    from blom_amd.gpu import BlomGpu, rccl_unique_id
    nsteps = 3
    case, masks, fields, ref = _single("chan_s", nsteps)
    t = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, case.nreg, masks)
    for nm, v in case.params.items():
        if not nm.endswith("0"):
            t.set(nm, v)
    put_fields(t, fields)
    t.set("delt1", case.params["baclin"])
    t.rccl_init(rccl_unique_id(), 0, 1)
    ref.step(0, nsteps)
    t.step(0, nsteps)
    t.sync()
    bad = [nm for nm in CHECK if not np.array_equal(ref.get(nm), t.get(nm))]
    t.rccl_finalize()
    t.close()
    ref.close()
    assert not bad, bad


@pytest.mark.parametrize("cfg,npx,npy,eddy", [("channel_tke", 2, 4, False), ("channel_tke", 2, 4, True),
                                              ("tnx2v1s_tke", 4, 2, False), ("tnx2v1s_tke", 4, 2, True)],
                         ids=["channel_2x4", "channel_2x4_eddy_fluxes", "tnx2v1s_4x2", "tnx2v1s_4x2_eddy_fluxes"])
def test_full_size_as_eight_tiles_reproduces_the_reference_checksums(cfg, npx, npy, eddy):
    """BASELINE.json configs 3 and 4 on one GPU.  The channel at full size (208x512x53, ntr = 3) cut into the 2 x 4 tiles
    of 104 x 128 that `bench.py --gpus 8` uses, and the tnx2v1 grid's dimensions (180x193x53, arctic patch) in the
    reference's own 4 x 2 tiling of bld/tnx2v1/patch.input.8 (tile columns of 45, tile rows of 97 and 96): the eight tiles
    on one device with the in-process transport (RCCL refuses several ranks per GPU).  After each of three steps the
    decomposition-independent checksum (xccrc chained over the tiles, blom_amd/tiles.py) of every recorded field must be
    the one the reference's own Fortran produced on ONE tile (tests/golden/<cfg>_crc.json; with `eddy` the fixture and the
    tiles hold the frozen non-zero eddy-induced mass fluxes of hostinit.frozen_eddy_fluxes, <cfg>_edf_crc.json)."""
    import json
    import os
    from blom_amd.gpu import BlomGpu, TileGroup
    from blom_amd import hostinit
    from blom_amd.checksum import grid_of
    from blom_amd.tiles import TileLayout, scatter_to_tile, chain_crc
    case = make_case(cfg)
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                       f"{cfg}_edf_crc.json" if eddy else f"{cfg}_crc.json")))
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    whole = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(whole, case)
    if eddy:
        hostinit.frozen_eddy_fluxes(whole, case)
    lay = TileLayout.regular(case.idm, case.jdm, npx, npy)
    if cfg.startswith("tnx2v1s"):
        assert lay.isizes == (45, 45, 45, 45) and lay.jsizes == (97, 96)      # bld/tnx2v1/patch.input.8
    grp = TileGroup(npx, npy)
    tiles = {}
    for py in range(npy):
        for px in range(npx):
            i0, j0, ii, jj = lay.tile(px, py)
            t = BlomGpu(ii, jj, case.kdm, case.ntr, nreg, {k: lay.window(masks[k], px, py) for k in masks},
                        itdm=case.idm, jtdm=case.jdm, i0=i0, j0=j0)
            for nm, v in case.params.items():
                if not nm.endswith("0"):
                    t.set(nm, v)
            scatter_to_tile(whole, t, lay, px, py)
            t.set("delt1", case.params["baclin"])
            t.set("eddtra_frozen", int(eddy))
            grp.attach(t, px, py)
            tiles[(px, py)] = t
    whole.close()
    names = [nm for nm in gold["crc"]["1"]["tmsmt2"] if eddy or nm not in ("umfltd", "vmfltd", "utfltd", "vtfltd", "usfltd", "vsfltd")]
    parts, errs = {}, []
    bar = threading.Barrier(npx * npy)

    def run(key):
        t = tiles[key]
        try:
            ns = 0
            for step in range(1, gold["nsteps"] + 1):
                ns = t.step(ns, 1)
                for nm in names:
                    if t.has_field(nm):
                        parts[(step, nm, key)] = t.crc_strips(nm, 1, t.field_info(nm)[0], grid_of(nm))
            t.sync()
        except Exception as e:
            errs.append(e)
            bar.abort()
    th = [threading.Thread(target=run, args=(k,), daemon=True) for k in tiles]
    [x.start() for x in th]
    [x.join(timeout=240) for x in th]
    assert not errs, errs
    assert not any(x.is_alive() for x in th), "a tile did not finish"
    bad, checked = [], 0
    for step in range(1, gold["nsteps"] + 1):
        for nm in names:
            if (step, nm, (0, 0)) not in parts:
                continue
            got = chain_crc({k: parts[(step, nm, k)] for k in tiles}, lay)
            want = gold["crc"][str(step)]["tmsmt2"][nm]
            checked += 1
            if got != want:
                bad.append(f"step {step} {nm}: 0x{got:08x} != 0x{want:08x}")
    for t in tiles.values():
        t.close()
    grp.destroy()
    assert not bad, bad[:20]
    assert checked >= 45


FORCING = ["ustarw", "swa", "nsf", "hmltfz", "lip", "sop", "eva", "rnf", "rfi", "fmltfz", "sfl", "swfc1", "swfc2", "swal1", "swal2", "ustar", "ustar3",
           "idkedt", "surflx", "sswflx", "surrlx", "salflx", "brnflx", "salrlx", "salt_corr", "sstclm", "ricclm", "sssclm", "uml", "vml", "umlres",
           "vmlres", "trflx", "trc_corr"]


DFE_IN = ["plat", "betatp", "cosang", "sinang", "hangle", "twedon", "ficem", "tdmls", "bdmlq"]


@pytest.mark.parametrize("cfg,npx,npy,live", [("chan_s_tke", 2, 2, False), ("tri_s_tke", 2, 2, False), ("box_s", 3, 1, False), ("tri_s_tke", 4, 2, False),
                                              ("chan_s_tke", 2, 2, True), ("tri_s_tke", 2, 2, True), ("box_s", 3, 1, True)])
def test_full_physics_step_on_tiles_matches_single_tile(cfg, npx, npy, live):
    """config 2's step as far as built (blomgpu_step with full_physics: + the front of difest_isobml, thermf with its two global
    sums, mxlayr, cmnfld1) on a decomposed domain: the sums are formed in the global domain's order from the owners' planes
    (halo.hip: xcsum_group), mxlayr's and the difest front's halo updates go through the tile transport.  A heat flux that
    changes sign across the domain and a fresh water flux drive entrainment and detrainment.  Interiors as on the single tile.
    live: with the diffusivity estimates of difest_isobml (stage_difest_iso.hip): kfil's and the diffusivities' halo updates through the
    tile transport, the neighbours' kmax from halo data."""
    from blom_amd.gpu import BlomGpu, TileGroup
    from blom_amd import hostinit
    nsteps = 4
    case, masks, fields, ref = _single(cfg, nsteps)
    kk = case.kdm
    hostinit.init_forcing(ref, case)
    nj, ni = case.jdm + 8, case.idm + 8
    y = np.linspace(-1.0, 1.0, nj)[:, None] + 0.0 * np.arange(ni)[None, :]
    x = np.linspace(0.0, 2 * np.pi, ni)[None, :] + 0.0 * y
    ref.put("nsf", (300.0 * (y + 0.3 * np.sin(x)))[None])
    ref.put("swa", (120.0 * (1.0 + np.cos(x)) * (y > -0.5))[None])
    ref.put("eva", (-2e-5 * (1.0 + 0.5 * np.sin(2 * x)))[None])
    ref.put("lip", (3e-5 * (y > 0.0))[None])
    area = ref.get_real("area") if hasattr(ref, "get_real") else None
    if live:
        hostinit.init_difest(ref, case, device=True)
    ii, jj = tile_extents(case, npx, npy)
    grp = TileGroup(npx, npy)
    tiles = {}
    for py in range(npy):
        for px in range(npx):
            tm = {k: tile_window(masks[k], case, npx, npy, px, py) for k in masks}
            t = BlomGpu(ii, jj, kk, case.ntr, case.nreg, tm, itdm=case.idm, jtdm=case.jdm, i0=px * ii, j0=py * jj)
            for nm, v in case.params.items():
                if not nm.endswith("0"):
                    t.set(nm, v)
            t.set("delt1", case.params["baclin"])
            grp.attach(t, px, py)
            tiles[(px, py)] = t
    scatter_state(ref, tiles, case, npx, npy, [f for f in ALL if f in fields] + [f for f in FORCING if ref.has_field(f)] + (DFE_IN if live else []))
    # the ocean area is a global number (mod_grid: area = xcsum(scp2, ips))
    scp2 = ref.get("scp2")[0][4:-4, 4:-4]
    w = masks["ip"][4:-4, 4:-4] > 0
    if case.nreg == 2:
        w = w.copy()
        w[-1, :] = False
    glob_area = float(np.sum(scp2[w]))

    def setup(g):
        g.set("area", glob_area)
        g.set("niwgf", 0.4)
        g.set("full_physics", 1)
        if live:
            import math
            for d_ in hostinit.DIFEST_NORESM:
                for nm_, v_ in d_.items():
                    if nm_ != "niwgf":
                        g.set(nm_, v_)
            g.set("bdml_logc", math.log(2. * hostinit._BVF0 / hostinit._CORI30))
            g.set("difest_live", 1)
    setup(ref)
    assert ref.step(0, nsteps) == nsteps
    errs = []

    def run(t):
        try:
            setup(t)
            assert t.step(0, nsteps) == nsteps
            t.sync()
        except Exception as e:          # a failing tile would leave the others at a barrier
            errs.append(e)
    th = [threading.Thread(target=run, args=(t,), daemon=True) for t in tiles.values()]
    [x_.start() for x_ in th]
    [x_.join(timeout=300) for x_ in th]
    assert not errs, errs
    bad = []
    for nm in CHECK + ["kfpla", "surflx", "salflx", "ustar", "mtkepe", "pbrnda", "sfl", "idkedt", "uml", "nslpx"] + (["difint", "difiso", "difdia", "difwgt", "L_scale"] if live else []):
        if not ref.has_field(nm):
            continue
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior(tiles, case, npx, npy, nm)
        wet = np.isfinite(a) & (np.abs(a) < 1e30)
        if not np.array_equal(a[wet], b[wet]):
            bad.append((nm, int((a[wet] != b[wet]).sum()), float(np.nanmax(np.abs(a[wet] - b[wet])))))
    pe = ref.get("mtkepe")[0, 4:-4, 4:-4]
    for t in tiles.values():
        t.close()
    ref.close()
    grp.destroy()
    assert not bad, bad
    assert (pe[masks["ip"][4:-4, 4:-4] > 0] != 0.0).any(), "mxlayr entrained nowhere"
