"""Kernel variants of the device library must agree bit for bit with each other: an optimised
kernel is the same arithmetic in a different storage/launch layout, so (unlike the comparison with
the CPU oracle, where libm's exp differs in the last place) no tolerance applies here."""
import numpy as np
import pytest

from blom_amd.cases import make_case
from blom_amd import hostinit
from parity import STATE_FIELDS

pytestmark = pytest.mark.gpu


def _run(cfg, nsteps, _extra=(), _ntr=None, **opts):
    from blom_amd.gpu import BlomGpu
    case = make_case(cfg, ntr=_ntr)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    for k, v in opts.items():
        gpu.set(k, v)
    hostinit.init_state(gpu, case)
    for k, v in opts.items():
        gpu.set(k, v)
    assert gpu.step(0, nsteps) == nsteps
    out = {nm: gpu.get(nm) for nm in STATE_FIELDS + list(_extra) if gpu.has_field(nm)}
    gpu.close()
    return out


@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 12), ("box_s", 8), ("fuk95", 6), ("chan_m", 6), ("tri_s", 8),
                                        ("chan_s_tke", 12), ("tri_s_tke", 8)])
@pytest.mark.parametrize("opt,variants", [("barotp_fused", (0, 1)), ("barotp_persist", (0, 1)), ("diapfl_du", (4, 8)),
                                          ("barotp_tile", (3216, 3208)), ("barotp_tile", (3216, 1608)), ("barotp_tile", (3216, 4016)), ("barotp_tile", (3216, 2616))])
def test_variants_bit_identical(cfg, nsteps, opt, variants):
    a = _run(cfg, nsteps, **{opt: variants[0]})
    b = _run(cfg, nsteps, **{opt: variants[1]})
    skip = {"util1", "util2", "util3", "util4"}        # scratch planes shared between stages
    bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
    detail = [(nm, int((~((a[nm] == b[nm]) | (np.isnan(a[nm]) & np.isnan(b[nm])))).sum())) for nm in bad]
    assert not bad, detail



@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 12), ("per_s", 8), ("fuk95", 6), ("chan_b", 8), ("chan_t8", 2), ("channel", 2)])
def test_four_barotropic_substeps_per_hand_off(cfg, nsteps):
    """k_bt_steps4 (temporal blocking, the default where the persistent form runs) against one odd+even pair per hand-off, against one
    kernel per equation, and against itself with a launch per four substeps: one tile, tiles with ragged last rows and columns across the
    periodic seam, walls in i (fuk95), 4 x 8 and 8 x 32 tiles at the sizes the bench runs."""
    from blom_amd.gpu import BlomGpu
    case = make_case(cfg)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    assert gpu.get_real("barotp_block_mode") == 1, "the blocked persistent form should serve this case by default"
    gpu.close()
    skip = {"util1", "util2", "util3", "util4"}
    a = _run(cfg, nsteps)
    for opts in (dict(barotp_block=0), dict(barotp_block=2), dict(barotp_fused=0, barotp_persist=0)):
        b = _run(cfg, nsteps, **opts)
        bad = [nm for nm in a if nm not in skip and a[nm].tobytes() != b[nm].tobytes()]
        assert not bad, (opts, bad)
    assert np.isfinite(a["ub"]).all() and np.abs(a["ub"]).max() > 0.0


@pytest.mark.parametrize("cfg,nsteps", [("chan_s_tke", 7), ("tri_s_tke", 6), ("box_s", 6)])
def test_round6_kernel_variants_bit_identical_down_to_the_sign_of_zero(cfg, nsteps):
    """The variants this round added or made the default -- pgforc's column kernel (pgf_uv_ring: 0 the kernel of rounds 1-5, 3 the
    double-buffered ring, 5 / 7 / 8 the lean form without / with the lazy reload / paired), the lean init_fluxes against the full one
    with the second stream off (advisor, round 5: the contract that only remap's storing tile kernel touches the flux arrays after
    the ring-only zeroing) -- compared as BYTES: a flux plane that kept last step's values, or a zero of the other sign, would show."""
    base = _run(cfg, nsteps)
    skip = {"util1", "util2", "util3", "util4"}
    for opts in (dict(pgf_uv_ring=0), dict(pgf_uv_ring=3), dict(pgf_uv_ring=5), dict(pgf_uv_ring=8), dict(pgf_uv_ring=0, pgf_reuse=1),
                 dict(lean_fluxes=0, overlap=0), dict(cmn_nslope_nb=2), dict(mom_aw_split=1), dict(mom_aw_split=2), dict(convec_nsingle=1000), dict(convec_nsingle=-1)):
        b = _run(cfg, nsteps, **opts)
        bad = [nm for nm in base if nm not in skip and base[nm].tobytes() != b[nm].tobytes()]
        assert not bad, (opts, bad)


@pytest.mark.parametrize("cfg,nsteps", [("chan_s_tke", 7), ("tri_s_tke", 6), ("fuk95", 5), ("box_s", 6)])
def test_tmsmt1_done_by_the_previous_steps_tmsmt2(cfg, nsteps):
    """blomgpu_step with several steps in one call: tmsmt2 also writes dpold, told, sold, trcold, dpuold, dpvold for the step
    that follows, which then launches no tmsmt1 (phy/mod_tmsmt.F90:230-277 copies exactly those values).  Against every step
    launching its own, and against one call per step; all fields, the *old ones included."""
    from blom_amd.gpu import BlomGpu
    old = ("told", "sold", "trcold")
    a = _run(cfg, nsteps, _extra=old, tmsmt_ahead=1)
    b = _run(cfg, nsteps, _extra=old, tmsmt_ahead=0)
    case = make_case(cfg)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    ns = 0
    for _ in range(nsteps):
        ns = gpu.step(ns, 1)
    d = {nm: gpu.get(nm) for nm in a}
    gpu.close()
    skip = {"util1", "util2", "util3", "util4"}
    for other, what in ((b, "own tmsmt1 every step"), (d, "one call per step")):
        bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], other[nm], equal_nan=True)]
        assert not bad, (what, bad)
    for nm in ("told", "sold", "dpold", "dpuold", "dpvold", "trcold"):
        assert nm in a, nm


@pytest.mark.parametrize("cfg,nsteps,ntr", [("chan_s_tke", 6, None), ("tri_s_tke", 5, None), ("box_s", 5, None), ("chan_s_tke", 4, 11),
                                            ("chan_m", 3, None)])
def test_remap_with_the_update_folded_in(cfg, nsteps, ntr):
    """option remap_fold: inside blomgpu_step the tile kernel of remap owns RT_TW-1 x RT_TH-1 cells of its tile, does
    k_remap_update's flux-divergence update for them (phy/mod_remap.F90:1468-1520) and hands dp, T, S and the advected tracers to
    pbcor1 through the work space; k_remap_ring commits what the update leaves in the halo.  Same bits in every array, halo
    included, as two kernels with the flux planes between them."""
    skip = {"util1", "util2", "util3", "util4"}
    kw = {} if ntr is None else {"_ntr": ntr}
    a = _run(cfg, nsteps, remap_fold=1, **kw)
    b = _run(cfg, nsteps, remap_fold=0, **kw)
    bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
    assert not bad, bad


@pytest.mark.parametrize("cfg,nsteps,ntr", [("chan_s_tke", 4, 11), ("tri_s_tke", 4, 16), ("box_s_tke", 3, 24)])
def test_remap_tracer_batches_whatever_rides_with_the_first_pass(cfg, nsteps, ntr):
    """more than four advected tracers: k_remap_tile evaluates the geometry once and streams the tracers through it in batches of
    four (two barriers a batch, stage_remap_tile.hip); option remap_nfirst = how many tracers ride with dp, T, S in the first
    pass instead.  Same bits for 0, 2 and 4 (the default), in every array."""
    skip = {"util1", "util2", "util3", "util4"}
    a = _run(cfg, nsteps, _ntr=ntr, remap_nfirst=0)
    for nf in (2, 4):
        b = _run(cfg, nsteps, _ntr=ntr, remap_nfirst=nf)
        bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
        assert not bad, (nf, bad)
    assert np.ptp(a["trc"][~np.isnan(a["trc"])]) > 0.0


@pytest.mark.parametrize("cfg", ["tri_m", "tri_m_tke"])
def test_persistent_barotp_with_the_arctic_patch(cfg):
    """the odd+even pairs of a barotropic phase in the persistent launch on a tripolar grid (tiles re-read rim AND seam row from
    their mirror tiles at the top of every pair) against one launch per pair and one kernel per equation"""
    a = _run(cfg, 8, barotp_arctic_persist=1)
    b = _run(cfg, 8, barotp_arctic_persist=0)
    d = _run(cfg, 8, barotp_fused=0, barotp_persist=0)
    skip = {"util1", "util2", "util3", "util4"}
    for other, what in ((b, "one launch per pair"), (d, "one kernel per equation")):
        bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], other[nm], equal_nan=True)]
        assert not bad, (what, bad)
    assert np.isfinite(a["ub"]).all() and float(np.abs(a["ub"]).max()) > 0.0


def test_variants_at_full_size_are_deterministic_and_identical():
    """BASELINE.json's channel size: hundreds of workgroups in flight, the regime where an
    inter-workgroup hazard (a tile updating a field in place while its neighbours still read the
    old values in their rims) shows up; small grids do not expose it."""
    keep = ("u", "v", "dp", "temp", "saln", "pb", "ub", "vb")
    runs = []
    for opts in ({"barotp_fused": 0, "barotp_persist": 0}, {}, {}):
        out = _run("channel", 2, **opts)
        runs.append({k: out[k] for k in keep})
    for nm in keep:
        assert np.array_equal(runs[1][nm], runs[2][nm], equal_nan=True), f"{nm}: not deterministic"
        assert np.array_equal(runs[0][nm], runs[1][nm], equal_nan=True), f"{nm}: production kernels != the one-kernel-per-equation barotp"


def _run_rccl_self(cfg, nsteps, **opts):
    """single rank whose periodic direction wraps onto itself through the RCCL transport"""
    from blom_amd.gpu import BlomGpu, rccl_unique_id
    case = make_case(cfg)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    for k, v in opts.items():
        gpu.set(k, v)
    gpu.rccl_init(rccl_unique_id(), 0, 1)
    assert gpu.step(0, nsteps) == nsteps
    gpu.sync()
    out = {nm: gpu.get(nm) for nm in STATE_FIELDS if gpu.has_field(nm)}
    gpu.rccl_finalize()
    gpu.close()
    return out


@pytest.mark.parametrize("cfg", ["tri_s", "tri_s_tke"])
def test_rccl_arctic_exchange_single_rank(cfg):
    """Arctic patch (nreg = 2) on the RCCL transport, one rank: E/W through send/recv to the rank itself, then the
    fold -- strip packed, with force = 1 sent to and received from the rank itself, fold targets filled from the
    gathered strips.  Must equal the single tile's halo rule (pinned on the reference built with ARCTIC)."""
    skip = {"util1", "util2", "util3", "util4"}
    a = _run(cfg, 6)
    assert all(np.isfinite(a[nm]).all() for nm in ("u", "dp", "temp"))
    for force in (0, 1):
        b = _run_rccl_2d_self(cfg, 6, force)
        bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
        assert not bad, (force, bad)


@pytest.mark.parametrize("overlap", [0, 1])
def test_rccl_exchange_overlapped_with_barotp_interior(overlap):
    """chan_m is three LDS tiles wide, so barotp's split launch (outer tile columns + exchange on the
    second stream, inner columns on the main stream) is active with barotp_overlap=1: same bits as the
    serial exchange and as the plain single tile"""
    skip = {"util1", "util2", "util3", "util4"}
    a = _run("chan_m", 6)
    b = _run_rccl_self("chan_m", 6, barotp_overlap=overlap)
    bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
    assert not bad, bad


def _run_rccl_2d_self(cfg, nsteps, force_ns):
    from blom_amd.gpu import BlomGpu, rccl_unique_id
    case = make_case(cfg)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    gpu.rccl_init_2d(rccl_unique_id(), 0, 1, 1)
    gpu.rccl_force_ns_exchange(force_ns)
    assert gpu.step(0, nsteps) == nsteps
    gpu.sync()
    out = {nm: gpu.get(nm) for nm in STATE_FIELDS if gpu.has_field(nm)}
    gpu.rccl_finalize()
    gpu.close()
    return out


@pytest.mark.parametrize("cfg,nsteps", [("fuk95", 3), ("per_s", 6)])
def test_rccl_north_south_exchange_single_rank(cfg, nsteps):
    """2-D RCCL transport on one GPU: in a j-periodic domain the N/S phase is forced through
    pack / ncclSend+Recv (to the rank itself) / unpack; per_s is periodic in i as well, so the corners
    then travel N/S first and E/W second, as between real tiles.  Must equal the plain halo update."""
    skip = {"util1", "util2", "util3", "util4"}
    a = _run(cfg, nsteps)
    assert all(np.isfinite(a[nm]).all() for nm in ("u", "dp", "temp"))
    for force in (0, 1):
        b = _run_rccl_2d_self(cfg, nsteps, force)
        bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
        assert not bad, (force, bad)


def test_full_size_mass_conservation():
    """BASELINE.json's channel size, 20 steps of the whole sequence: the flux-form continuity equations
    (advect, eddtra, barotp, pbcor) conserve the global mass integral to rounding (both as the sum of the
    layer thicknesses and as the barotropic bottom pressure), and the state stays finite"""
    from blom_amd.gpu import BlomGpu
    case = make_case("channel", nslp0=2e-4)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    kk, J, I = case.kdm, slice(4, 4 + case.jdm), slice(4, 4 + case.idm)
    wet = ip[J, I] > 0
    area = gpu.get("scp2")[0][J, I] * wet

    def mass(n):
        nn = (n - 1) * kk
        pb = gpu.get("pb")[n - 1][J, I]
        dps = gpu.get("dp")[nn:nn + kk, J, I].sum(0)
        return float((pb * area).sum()), float((dps * area).sum()), float(np.abs(dps - pb)[wet].max() / pb[wet].max())

    ns = gpu.step(0, 1)
    m0 = mass(ns % 2 + 1)
    for _ in range(20):
        ns = gpu.step(ns, 1)
        m = mass(ns % 2 + 1)
        assert abs(m[0] - m0[0]) <= 1e-13 * abs(m0[0]) and abs(m[1] - m0[1]) <= 1e-13 * abs(m0[1]), (ns, m, m0)
    for nm in ("u", "v", "dp", "temp", "saln"):
        assert np.isfinite(gpu.get(nm)).all(), nm
    gpu.close()


def test_tracer_switch_that_is_not_built_fails_loudly():
    """cppm advects every tracer (phy/mod_cppm.F90 has no TKEADV switch), so `TKE tracers not advected` together with
    advmth = 'cppm' is a combination the reference cannot be in; the library says so instead of guessing"""
    from blom_amd.gpu import BlomGpu
    case = make_case("chan_s_tk0", advmth="cppm")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    case.params["tkeadv"] = 1
    hostinit.init_state(gpu, case)
    gpu.set("tkeadv", 0)
    with pytest.raises(Exception, match="TKEADV"):
        gpu.step(0, 1)
    gpu.set("tkeadv", 1)
    assert gpu.step(0, 1) == 1
    gpu.close()


def test_graphs_are_captured_inside_a_warm_up_of_three_steps():
    """Option use_graph: both parities are captured in the third step after the last change of an option or of a parameter in the device's
    view -- the time step changes after the first step of a run from rest, so its fourth step is the first replay -- and the stage timers
    do not drop them.  A capture that silently failed would leave the step on plain launches."""
    from blom_amd.gpu import BlomGpu
    case = make_case("chan_s")
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    gpu.set("use_graph", 1)                                # (off by default: replay measures slower than plain launches)
    ns = gpu.step(0, 4)
    assert gpu.get_real("graph_steps") == 1 and gpu.get_real("graph_failures") == 0      # the capturing step itself is a replay
    ns = gpu.step(ns, 5)
    assert gpu.get_real("graph_steps") == 6
    gpu.set("timing", 1)                                   # plain launches beside the graphs ...
    ns = gpu.step(ns, 2)
    assert gpu.get_real("graph_steps") == 6
    gpu.set("timing", 0)                                   # ... which are still there
    ns = gpu.step(ns, 2)
    assert gpu.get_real("graph_steps") == 8 and gpu.get_real("graph_failures") == 0
    gpu.close()


@pytest.mark.parametrize("cfg,nsteps", [("chan_s", 12), ("tri_s", 10), ("fuk95", 8)])
def test_step_replayed_as_a_hip_graph_is_identical(cfg, nsteps):
    """blomgpu_step captures the stage sequence of a step for both parities of the time levels at once (after two plain steps) and
    replays it; the persistent barotp kernel's epochs restart with every barotp call so that the launches are the same
    from step to step.  Same bits as plain launches."""
    a = _run(cfg, nsteps, use_graph=0)
    b = _run(cfg, nsteps, use_graph=1)
    skip = {"util1", "util2", "util3", "util4"}
    bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
    assert not bad, bad


@pytest.mark.parametrize("cfg", ["chan_m", "tri_m"])
def test_halo_exchange_overlapped_with_the_inner_tiles_of_remap(cfg):
    """RCCL transport (one rank sending to itself): the exchange of cau, cav and the tracers in front of remap on the second
    stream while k_remap_tile runs the tiles that read no halo point, the edge tiles afterwards (halo_overlap = 1, the default
    of the RCCL transport): same bits as the serial exchange and as the plain single tile.  chan_m is 80 x 40: 3 x 5 tiles of
    32 x 8, three of them inner; tri_m (64 x 40) has the arctic patch: the exchange rewrites the seam row, so the tiles whose rim
    reaches it wait as well."""
    skip = {"util1", "util2", "util3", "util4"}
    a = _run(cfg, 6)
    for ovl in (0, 1):
        b = _run_rccl_self(cfg, 6, halo_overlap=ovl)
        bad = [nm for nm in a if nm not in skip and not np.array_equal(a[nm], b[nm], equal_nan=True)]
        assert not bad, (ovl, bad)


def test_variants_identical_on_tnx2v1s():
    """the tripolar grid at tnx2v1's size (180 x 193 x 53, arctic patch, unequal wet areas): this round's kernels against
    round 1's, and the pair kernel's seam rule at the tile load against the separate halo launches of the unfused path"""
    keep = ("u", "v", "dp", "temp", "saln", "pb", "ub", "vb", "trc")
    old = _run("tnx2v1s", 3, barotp_fused=0, barotp_persist=0)
    new = _run("tnx2v1s", 3)
    for nm in keep:
        assert np.array_equal(old[nm], new[nm], equal_nan=True), nm


@pytest.mark.parametrize("cfg,nsteps,live", [("chan_s_tke", 9, True), ("tri_s_tke", 7, True), ("channel_tke", 4, True), ("chan_s_tke", 7, False)])
def test_physics_stages_side_by_side_on_the_second_stream(cfg, nsteps, live):
    """Round 6, option phys_dag (bits 1 / 2 / 4 / 8): inside blomgpu_step's full step with live slopes and diffusivities cmnfld2's column
    kernels run on the second stream beside difest_isobml's common part and vertical chain, difest's lateral part behind them there;
    diapfl's momentum mixing beside thermf and mxlayr's first kernels; updtrc's ideal-age step beside barotp's first kernels; mxlayr's copy-back clamp beside the rest of mxlayr (not with the arctic patch).  The
    stages share no array one side writes (stage_cmnfld.hip: st_cmnfld2, stage_difest_iso.hip: st_difest_isobml, stage_diapfl.hip:
    st_diapfl) -- a missed dependency is a race and shows as differing BYTES against the one-stream order (phys_dag = 0, overlap = 0),
    at BASELINE.json's channel size too, where thousands of wavefronts of both sides are in flight, and from run to run."""
    import bench
    case = make_case(cfg)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    masks = dict(ip=ip, iu=iu, iv=iv, iq=iq)
    names = STATE_FIELDS + ["nslpx", "nslpy", "nnslpx", "nnslpy", "bfsqf", "ustar", "surflx"]

    def run(**opts):
        gpu = bench.device_for_bench(case, nreg, masks, live=live)      # (live = False: frozen diffusivities, blomgpu_step joins cmnfld2's kernels itself)
        for k, v in opts.items():
            gpu.set(k, v)
        assert gpu.step(0, nsteps) == nsteps
        out = {nm: gpu.get(nm).tobytes() for nm in names if gpu.has_field(nm)}
        gpu.close()
        return out

    base = run(phys_dag=0, overlap=0)
    # (use_graph: the forks and joins of the second stream inside a captured step)
    for opts in (dict(phys_dag=15), dict(phys_dag=15), dict(phys_dag=1), dict(phys_dag=14), dict(phys_dag=7), dict(phys_dag=0), dict(phys_dag=15, use_graph=1)):
        b = run(**opts)
        bad = [nm for nm in base if base[nm] != b[nm]]
        assert not bad, (opts, bad)
