"""The library's RCCL transport with SEVERAL ranks, on the CPU: tests/hostemu compiles the device library's own
sources for the host (kernels run thread by thread, one fiber per thread) and stands in for RCCL with mailboxes between
host threads, so every rank here is a thread with its own context running the real pack / ncclSend+Recv / unpack code
of blom_amd/csrc/comm_rccl.hip -- tile grids with several rows and columns, tiles of unequal height as in the
reference's bld/tnx2v1/patch.input.8, closed and periodic and tripolar domains.  Requirement: interiors bit-identical to
the single tile, and the decomposition-independent checksum (xccrc) chained over the tiles equal to the single tile's.
(Several ranks cannot share the one GPU of the test box -- RCCL refuses it -- so this is where that code runs.)"""
import os
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu", "libblomgpu_hostemu.so")
pytestmark = pytest.mark.skipif(not os.path.exists(EMU), reason="tests/hostemu/libblomgpu_hostemu.so not built")

CHECK = ["u", "v", "dp", "temp", "saln", "sigma", "pb", "ub", "vb", "ubflxs_p", "pb_p", "trc", "uflx", "vflx",
         "pgfx", "pgfy", "dpu", "dpv", "pbu", "pbv", "ubflx", "vbflx", "pb_mn", "ubcors_p"]


@pytest.fixture()
def emu_lib():
    import blom_amd.gpu as g
    old = g.LIB_PATH
    g.LIB_PATH = EMU
    yield
    g.LIB_PATH = old


def _run_case(cfg, isizes, jsizes, nsteps=3, bt_global=False, cppm=False, hybrid=None, full=False):
    from blom_amd.gpu import BlomGpu, rccl_unique_id
    from blom_amd.tiles import TileLayout, scatter_to_tile, gather_interior_layout, chain_crc, make_barotp_global
    from test_gpu_tiles import _single
    if cppm:
        from blom_amd import hostinit
        from blom_amd.cases import make_case
        case = make_case(cfg, advmth="cppm")
        nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
        masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
        ref = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
        hostinit.init_state(ref, case)
        ref.set("delt1", case.params["baclin"])
    else:
        case, masks, fields, ref = _single(cfg, nsteps)
    lay = TileLayout(tuple(isizes), tuple(jsizes))
    assert lay.itdm == case.idm and lay.jtdm == case.jdm
    check = list(CHECK)
    if hybrid:                                            # (vcoord_type, regrid_method): the step of the hybrid coordinate, DESIGN.md 3h
        from test_gpu_tiles import hybrid_inputs
        from blom_amd.hostinit import step_indices
        for nm, a in hybrid_inputs(case).items():
            ref.put(nm, a)
        kk = case.kdm
        pbot = float(np.max(ref.get("p")[kk][masks["ip"] > 0]))
        plevel = 0.3 * pbot * (np.arange(kk) / kk) ** 1.3
        check += ["umfltd", "vmfltd", "umflsm", "vmflsm", "hml_tf", "mld", "bfsqi", "nslpx", "nslpy", "salt_corr", "buoyfl"]

        def setup_hybrid(g):
            g.set("vcoord_type", hybrid[0])
            g.set("ale_regrid_method", hybrid[1])
            g.set("mlrmth", "fox08")
            g.set("swamxd", 200.0)
            g.set("brine_mlbase_frac", 0.4)
            g.set_vector("plevel", plevel)
            if len(hybrid) > 2 and hybrid[2]:                 # ltedtp = 'neutral': neutral diffusion inside ale_regrid_remap
                g.set("ltedtp_opt", 2)
                g.set("ndiff_surface_align", 1)
            g.stage("cmnfld1", *step_indices(0, kk))
    extra_fields = []
    if full:                                              # config 2's step as far as built (thermf's global sums, mxlayr, difest front)
        from test_gpu_tiles import FORCING
        from blom_amd import hostinit as hostinit_mod
        hostinit_mod.init_forcing(ref, case)
        nj, ni = case.jdm + 8, case.idm + 8
        yy = np.linspace(-1.0, 1.0, nj)[:, None] + 0.0 * np.arange(ni)[None, :]
        ref.put("nsf", (300.0 * yy)[None])
        ref.put("swa", (120.0 * (yy > -0.5))[None])
        ref.put("eva", (-2e-5 * np.ones((nj, ni)))[None])
        extra_fields = [f for f in FORCING if ref.has_field(f)]
        scp2 = ref.get("scp2")[0][4:-4, 4:-4]
        w = masks["ip"][4:-4, 4:-4] > 0
        if case.nreg == 2:
            w = w.copy()
            w[-1, :] = False
        glob_area = float(np.sum(scp2[w]))
        check += ["kfpla", "surflx", "ustar", "mtkepe", "sfl", "uml"]

        def setup_full(g):
            g.set("area", glob_area)
            g.set("niwgf", 0.4)
            g.set("full_physics", 1)
    uid = rccl_unique_id()
    tiles, errs, crcs = {}, [], {}
    crc_fields = [("dp", 1, 2 * case.kdm, 1), ("u", 1, 2 * case.kdm, 3), ("v", 1, 2 * case.kdm, 4), ("pb", 1, 2, 1)]
    lock = threading.Lock()

    def rank_main(rank):
        try:
            px, py = lay.rank_tile(rank)
            i0, j0, ii, jj = lay.tile(px, py)
            tm = {k: lay.window(masks[k], px, py) for k in masks}
            t = BlomGpu(ii, jj, case.kdm, case.ntr, case.nreg, tm, itdm=case.idm, jtdm=case.jdm, i0=i0, j0=j0)
            for nm, v in case.params.items():
                if not nm.endswith("0"):
                    t.set(nm, v)
            t.set("delt1", case.params["baclin"])
            with lock:                                    # the whole-domain backend is shared: read it one rank at a time
                scatter_to_tile(ref, t, lay, px, py)
                glob = make_barotp_global(ref, case, masks) if bt_global else None
            t.rccl_init_2d(uid, rank, lay.npx, lay.npy)
            if bt_global:                                 # every rank solves the whole 2-D barotropic domain
                t.rccl_attach_barotp_global(glob, lay.isizes, lay.jsizes)
            with lock:
                tiles[(px, py)] = t
            barrier.wait()
            if cppm:                                      # every rank builds its coefficient tables (halo updates included)
                t.stage("init_cppm", 2, 1, case.kdm, 0, case.kdm + 1, 1)
            if hybrid:
                setup_hybrid(t)
            if full:
                setup_full(t)
            assert t.step(0, nsteps) == nsteps
            t.sync()
            crcs[(px, py)] = {f[0]: t.crc_strips(*f) for f in crc_fields}
        except Exception as e:                            # a failing rank would leave the others waiting
            errs.append(e)
            barrier.abort()

    n = lay.npx * lay.npy
    barrier = threading.Barrier(n)
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
    [x.start() for x in th]
    [x.join(timeout=600) for x in th]
    assert not errs, errs
    if cppm and hybrid:
        ref.stage("init_cppm", 2, 1, case.kdm, 0, case.kdm + 1, 1)
    if hybrid:
        setup_hybrid(ref)
    if full:
        setup_full(ref)
    assert ref.step(0, nsteps) == nsteps
    bad = []
    for nm in check:
        a = ref.get(nm)[:, 4:4 + case.jdm, 4:4 + case.idm]
        b = gather_interior_layout(tiles, lay, nm)
        ok = np.isfinite(a) & (np.abs(a) < 1e30)          # land keeps whatever pattern it had
        if not np.array_equal(a[ok], b[ok]):
            bad.append((nm, int((a[ok] != b[ok]).sum())))
    crc_bad = []
    for f in crc_fields:
        want = ref.crc(*f)
        got = chain_crc({k: v[f[0]] for k, v in crcs.items()}, lay)
        if want != got:
            crc_bad.append((f[0], hex(want), hex(got)))
    for t in tiles.values():
        t.rccl_finalize()
        t.close()
    ref.close()
    assert not bad, bad
    assert not crc_bad, crc_bad


@pytest.mark.parametrize("cfg,isizes,jsizes", [
    ("chan_s", (10, 10), (12, 12)),            # periodic in i, closed in j: 2 x 2
    ("chan_s", (20,), (13, 11)),               # tile rows of unequal height
    ("chan_s", (7, 7, 6), (24,)),              # three tile columns of unequal width, E/W wrap between first and last
    ("box_s", (12, 12), (11, 9)),              # closed basin with an island, unequal rows
    ("tri_s", (12, 12), (11, 9)),              # arctic patch: fold across the top row between mirror tiles
    ("tri_s_tke", (6, 6, 6, 6), (10, 10)),     # the reference's 4 x 2 shape of bld/tnx2v1/patch.input.8
])
def test_rccl_ranks_match_single_tile(emu_lib, cfg, isizes, jsizes):
    _run_case(cfg, isizes, jsizes)


@pytest.mark.parametrize("cfg,isizes,jsizes,vcoord,method,cppm", [
    ("chan_s", (10, 10), (13, 11), "cntiso_hybrid", "nudge", False),
    ("box_s", (12, 12), (11, 9), "cntiso_hybrid", "direct", True),
    ("tri_s", (12, 12), (11, 9), "cntiso_hybrid", "nudge", False),
    ("tri_s", (6, 6, 6, 6), (10, 10), "plevel", "direct", True),
])
def test_rccl_ranks_with_the_hybrid_step(emu_lib, cfg, isizes, jsizes, vcoord, method, cppm):
    """the step of the hybrid vertical coordinate (ale_regrid_remap with its smoothing ring, eddtra_ale, ale_vdiffm's viscosity
    halo, the hybrid branches of cmnfld) through the RCCL transport, tripolar grids included"""
    _run_case(cfg, isizes, jsizes, cppm=cppm, hybrid=(vcoord, method))


@pytest.mark.parametrize("cfg,isizes,jsizes", [("chan_s_tke", (10, 10), (13, 11)), ("tri_s_tke", (6, 6, 6, 6), (10, 10)), ("box_s", (12, 12), (11, 9))])
def test_rccl_ranks_with_the_full_physics_step(emu_lib, cfg, isizes, jsizes):
    """config 2's step as far as built through the RCCL transport: thermf's two global sums travel like the barotropic solver's
    planes and are formed on the replicated solve's global context (comm_rccl.hip: rccl_xcsum_dev); mxlayr, the difest front"""
    _run_case(cfg, isizes, jsizes, bt_global=True, full=True)


@pytest.mark.parametrize("cfg,isizes,jsizes,method", [("chan_s", (10, 10), (13, 11), "nudge"), ("tri_s", (12, 12), (11, 9), "nudge"),
                                                      ("box_s", (12, 12), (11, 9), "direct")])
def test_rccl_ranks_with_neutral_diffusion(emu_lib, cfg, isizes, jsizes, method):
    """the hybrid step with ltedtp = 'neutral' through the RCCL transport: the regridding two rings beyond the tile, the face
    searches on the tile's edge, the slopes' halo update of cmnfld_nnslope_ale"""
    _run_case(cfg, isizes, jsizes, hybrid=("cntiso_hybrid", method, True))


@pytest.mark.parametrize("cfg,isizes,jsizes", [
    ("chan_s", (7, 7, 6), (13, 11)),
    ("box_s", (12, 12), (11, 9)),
    ("tri_s", (12, 12), (11, 9)),              # arctic patch: tags and edge coefficients swap across the seam (mod_cppm.F90:2650-2722)
    ("tri_s_tke", (6, 6, 6, 6), (10, 10)),
])
def test_rccl_ranks_with_cppm(emu_lib, cfg, isizes, jsizes):
    """advmth = 'cppm' through the RCCL transport, also on a decomposed tripolar grid"""
    _run_case(cfg, isizes, jsizes, cppm=True)


@pytest.mark.parametrize("cfg,isizes,jsizes", [
    ("chan_s", (10, 10), (12, 12)),
    ("chan_s", (7, 7, 6), (13, 11)),           # unequal columns and rows
    ("box_s", (12, 12), (11, 9)),
    ("tri_s", (12, 12), (11, 9)),              # arctic patch: the global context folds onto itself, no strips travel for barotp
    ("tri_s_tke", (6, 6, 6, 6), (10, 10)),
])
def test_rccl_ranks_with_replicated_barotropic_solve(emu_lib, cfg, isizes, jsizes):
    """blomgpu_rccl_attach_barotp_global: the tiles gather the 2-D fields barotp reads once per step, every rank solves the
    whole barotropic domain on a second context and takes its window -- no exchange inside the substep loop.  Same
    requirement: the single tile's bits."""
    _run_case(cfg, isizes, jsizes, bt_global=True)


def test_patch_input_layout():
    from blom_amd.tiles import TileLayout
    p = "/root/reference/bld/tnx2v1/patch.input.8"
    if not os.path.exists(p):
        pytest.skip("reference tree not present")
    lay = TileLayout.from_patch_input(p)
    assert lay.isizes == (45, 45, 45, 45) and lay.jsizes == (97, 96)
    assert TileLayout.regular(180, 193, 4, 2).jsizes == (97, 96)
    assert lay.tile(3, 1) == (135, 97, 45, 96)
