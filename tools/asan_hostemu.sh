#!/bin/bash
# TEST INFRASTRUCTURE: the device library's kernels, compiled for the host with AddressSanitizer (make -C tests/hostemu asan),
# stepped through whole stage sequences -- small grids, 6 and 11 tracers, cppm, the arctic patch, a barotropic solve with one
# kernel per equation, the step of the hybrid vertical coordinate (ale_regrid_remap, ale_forcing, ale_vdifft/m, ..), and 3-8 emulated
# RCCL ranks with the decomposed and the replicated barotropic solve.  GPU address
# sanitizer runs are not available on the pool; this is the CPU-side check.  Prints one line per case; any ASan report fails it.
cd "$(dirname "$0")/.." || exit 1
make -C tests/hostemu asan -j8 > /dev/null || exit 1
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
export BLOMGPU_LIB=$PWD/tests/hostemu/build_asan/libblomgpu_hostemu_asan.so
python3 - <<'PY' 2>&1 | grep -v "tracer count"
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from blom_amd.cases import make_case
from blom_amd import hostinit
from blom_amd.gpu import BlomGpu
def run(cfg, steps=3, ntr=None, **opts):
    over = {"advmth": opts.pop("advmth")} if "advmth" in opts else {}
    case = make_case(cfg, ntr=ntr, **over)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    gpu.set("live_slopes", 1)
    for k, v in opts.items():
        gpu.set(k, v)
    ns = gpu.step(0, steps)
    assert np.isfinite(gpu.get("u")).all()
    print(cfg, ntr, over, opts, "ok", ns, flush=True)
    gpu.close()
for cfg in ("chan_s_tke", "tri_s_tke", "box_s", "fuk95"):
    run(cfg, 4)
run("chan_s_tke", ntr=11)
run("tri_s", ntr=6)
run("fuk95", advmth="cppm")
run("tri_s", advmth="cppm")
run("chan_m", steps=2, barotp_fused=0)
run("tri_m", steps=2)
# the step of the hybrid vertical coordinate (ale_regrid_remap with both coordinates and regrid methods, ale_forcing, ale_vdifft/m ..)
def hybrid(cfg, vcoord, method, advmth="remap", steps=3, ntr=None, neutral=False):
    case = make_case(cfg, advmth=advmth, ntr=ntr)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    z = np.arange(kk + 1)[:, None, None] / kk
    frac = np.clip(1.0 - z / 0.4, 0.0, 1.0) ** 2 * np.ones((1, nj, ni))
    for nm in ("kvisc_m", "kdiff_t", "kdiff_s"):
        gpu.put(nm, 1e-4 * np.ones((kk + 1, nj, ni)))
    for nm in ("t_ns_nonloc", "s_nb_nonloc", "t_rs_nonloc", "s_rs_nonloc", "mu_nonloc", "mv_nonloc"):
        gpu.put(nm, frac)
    for nm, v in (("swfc1", .6), ("swfc2", .4), ("swal1", 1.), ("swal2", 15.), ("surflx", -50.), ("sswflx", -80.), ("salflx", 1e-3), ("OBLdepth", 30.)):
        gpu.put(nm, v * np.ones((1, nj, ni)))
    pbot = float(np.max(gpu.get("p")[kk][4:-4, 4:-4][ip[4:-4, 4:-4] > 0]))
    gpu.set("vcoord_type", vcoord)
    gpu.set("ale_regrid_method", method)
    gpu.set("mlrmth", "fox08")
    gpu.set_vector("plevel", 0.3 * pbot * (np.arange(kk) / kk) ** 1.3)
    if neutral:                                         # neutral diffusion inside ale_regrid_remap (stage_ndiff.hip)
        gpu.set("ltedtp_opt", 2)
        gpu.set("ndiff_surface_align", 1)
    if advmth == "cppm":
        gpu.stage("init_cppm", 2, 1, kk, 0, kk + 1, 1)
    gpu.stage("cmnfld1", *hostinit.init_indices(0, kk))
    ns = gpu.step(0, steps)
    u = gpu.get("u")[:, 4:-4, 4:-4]
    assert np.isfinite(u[np.broadcast_to((iu[4:-4, 4:-4] > 0)[None], u.shape)]).all()
    print(cfg, vcoord, method, advmth, ntr, "neutral" if neutral else "layer", "hybrid step ok", ns, flush=True)
    gpu.close()
hybrid("chan_s", "cntiso_hybrid", "nudge")
hybrid("tri_s", "cntiso_hybrid", "direct")
hybrid("box_s", "plevel", "direct", advmth="cppm")
hybrid("fuk95", "plevel", "nudge", advmth="cppm", steps=2)
hybrid("chan_s_tke", "cntiso_hybrid", "nudge", ntr=9)            # further passes of the fused vdifft kernel, two engine batches
hybrid("tri_s_tke", "cntiso_hybrid", "direct", ntr=6)
hybrid("chan_s", "cntiso_hybrid", "nudge", neutral=True)
hybrid("tri_s", "cntiso_hybrid", "nudge", neutral=True)
hybrid("box_s", "plevel", "direct", advmth="cppm", neutral=True)
hybrid("chan_s_tke", "cntiso_hybrid", "nudge", ntr=9, neutral=True)   # two engine batches of coefficients, nine fluxes per record
# config 2's step as far as built: thermf, mxlayr, the front of difest_isobml, cmnfld1 (full_physics)
def full(cfg, steps=4, ntr=None):
    case = make_case(cfg, ntr=ntr)
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq))
    hostinit.init_state(gpu, case)
    hostinit.init_forcing(gpu, case)
    nj, ni = case.jdm + 8, case.idm + 8
    y = np.linspace(-1.0, 1.0, nj)[:, None] + 0.0 * np.arange(ni)[None, :]
    gpu.put("nsf", (300.0 * y)[None])
    gpu.put("swa", (120.0 * (y > -0.5))[None])
    gpu.put("eva", (-2e-5 * np.ones((nj, ni)))[None])
    gpu.set("full_physics", 1)
    ns = gpu.step(0, steps)
    assert np.isfinite(gpu.get("u")[:, 4:-4, 4:-4]).all()
    print(cfg, ntr, "full physics ok", ns, flush=True)
    gpu.close()
full("chan_s_tke")
full("tri_s_tke")
full("box_s", ntr=5)
import test_hostemu_multirank as t
for cfg, isz, jsz in (("chan_s", (7, 7, 6), (13, 11)), ("tri_s_tke", (6, 6, 6, 6), (10, 10))):
    for g in (True, False):
        t._run_case(cfg, isz, jsz, bt_global=g)
        print(cfg, isz, jsz, "replicated" if g else "decomposed", "barotropic solve ok", flush=True)
PY
exit ${PIPESTATUS[0]}
