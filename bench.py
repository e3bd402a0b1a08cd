#!/usr/bin/env python3
"""Benchmark of the MI355X-native BLOM dynamical core.

  python bench.py --gpus N --steps K --warmup W

A "step" is one baroclinic time step of the hot path (stage sequence of
phy/mod_blom_step.F90:96-253 restricted to the dynamical core: init_fluxes, tmsmt1, [halo updates
of cmnfld2/difest], eddtra, advect(remap), pbcor1, diffus, pgforc, momtum, convec, diapfl, [mxlayr
dp-halo tail], updtrc, barotp, pbcor2, tmsmt2) on the `channel` configuration of BASELINE.json
(208x512x53), with the state resident in HBM.  metric = simulated model days per wall second
= steps/s * baclin / 86400 of the domain that is integrated.

N > 1 (default --scaling strong): BASELINE.json's configs 3/4 -- the SAME domain cut into npx x npy tiles
in the reference's patch.input scheme (channel: 1x2, 2x2, 2x4 tiles; tnx2v1s: 2x1, 2x2 and the
reference's 4x2 with tile rows of 97 and 96), one tile per GPU, halos over RCCL send/recv.  The
checksum of the final state (xccrc chained over the tiles) is decomposition independent: it is the
same for every N at equal --steps/--warmup.  --scaling weak keeps round 1's mode (the channel made
N times as long, one 208x512 tile per GPU).  Launched without torchrun (`python bench.py --gpus N`),
the script starts the N ranks itself.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, algorithmic bytes /
HIP-event duration vs 8 TB/s), `step_roofline` (SURVEY.md 8d A_step / step time), `stages`
(ms per stage) and `cpu_baseline` (the reference's own compiled Fortran, or the C restatement,
timed on a bounded number of steps on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
NSLP0 = 2.0e-4             # amplitude of the frozen isopycnal slopes that drive eddtra (cases.py)


FORCING = {"default": {}, "calm": {"ustarw0": 5.0e-5}}


def build_case(cfg, advmth="remap", tracers="default", forcing="default"):
    import numpy as np
    from blom_amd.cases import make_case
    from blom_amd import hostinit
    # "default": the reference's default build options (meson_options.txt:17-21: TKE + advection of it, ideal age)
    # => ntr = 3, the tracer count SURVEY.md 8(d) quotes the channel on; "iage": -DTRC -DIDLAGE only, ntr = 1
    # an integer N > 3: the default set plus N - 3 passive tracers (what iHAMOCC's are to the dynamical core, config 5)
    ntr = int(tracers) if tracers.isdigit() else None
    # forcing: "default" = channel/mod_channel.F90:365 (ustarw = 0.005, which thermf_channel's 1e2 turns into 0.5 m/s of friction
    # velocity); "calm" = this project's second published set, ustarw = 5e-5 (0.005 m/s after the factor)
    case = make_case(cfg + ("" if tracers == "iage" else "_tke"), ntr=ntr, nslp0=NSLP0, advmth=advmth, **FORCING[forcing])
    nreg, _, ip, iu, iv, iq = hostinit.bigrid_np(case.depth, case.idm, case.jdm, arctic=case.nreg == 2)
    return case, nreg, dict(ip=ip, iu=iu, iv=iv, iq=iq)


def ntr_diffused(case):
    """Tracers diffus acts on: the TKE and length-scale tracers are left out unless built with -DTKEIDF
    (phy/mod_diffus.F90:64-66)."""
    p = case.params
    return case.ntr - (2 if p.get("itrtke", -1) >= 1 and not p.get("tkeidf", 0) else 0)


def algorithmic_bytes(case, ntr):
    """A_step = (124 + 6 ntr) F + 2.5 lstep 62 G; of the 6 ntr, 2 ntr are diffus' and count
    only the tracers diffus acts on."""
    F = case.idm * case.jdm * case.kdm * 8.0
    G = case.idm * case.jdm * 8.0
    a3d = (124 + 4 * ntr + 2 * ntr_diffused(case)) * F
    a2d = 2.5 * case.params["lstep"] * 62 * G
    return a3d, a2d


# compulsory HBM bytes per launch of each timed kernel class, in units of F (one 3-D field):
# every distinct array the class reads or writes counted once (2-D coefficient arrays ignored);
# see DESIGN.md "Kernels" for the derivation.
def class_bytes_F(ntr, ntr_dif=None):
    ntr_dif = ntr if ntr_dif is None else ntr_dif
    return {
        "remap": 25 + 2 * ntr, "diffus": 19 + 2 * ntr_dif, "pgforc": 15, "momtum": 26, "eddtra": 14,
        "cppm": 48 + 4 * ntr, "diapfl": 23 + 2 * ntr, "pbcor1": 12 + 2 * ntr, "pbcor2": 13 + 2 * ntr, "convec": 19 + 2 * ntr,
    }


def kernel_bytes_F(ntr, nadv):
    """Algorithmic bytes of the kernels that can be the step's longest one, in units of F: every distinct array the KERNEL reads or
    writes counted once (its scratch planes are outputs / inputs like any other: they are what the kernel is asked to move).
      k_remap_tile      R dp, T, S, nadv tracers (time level n), cau, cav; W the six mass / heat / salt flux planes + 2 per tracer
      k_mom_cor_march   R dp, u, v (m), u, v (n), dpu, dpv, p, pgfx, pgfy (m, n, old), dpuold, dpvold, visu, visv;
                        W the four updated-velocity planes, absvor, dpvor
      k_mom_visc_march  R u, v (n), pu, pv, dpu, dpv; W visu, visv
      k_diapfl_column3  R T, S, dp, sigma, tracers, sigmar, difdia; W T, S, dp, sigma, tracers, p, fpug, fplg, difdia"""
    return {"k_remap_tile": (3 + nadv + 2) + (6 + 2 * nadv), "k_mom_cor_march": 18 + 6, "k_mom_visc_march": 6 + 2,
            "k_diapfl_column3": (6 + ntr) + (8 + ntr),
            # k_pgf_uv   R p, T, S (n), phi, phi', pu, pv, dpu, dpv, pgfx, pgfy (the values that become the _o copies); W pgfx, pgfy, pgfx_o, pgfy_o
            # k_pbc_tile R dp, S, T, tracers, the six flux planes; W the six flux planes, the new dp, S, T, tracers
            "k_pgf_uv": 11 + 4, "k_pbc_tile": (3 + ntr + 6) + (6 + 3 + ntr)}


KNOWN_CONFIGS = ("channel", "chandyn", "chanovl0", "tnx2v1s", "tnx1v4s", "chan_t8", "hybrid", "hor3map", "ale")   # chandyn: the channel with --physics dyncore; chanovl0: with --opt overlap=0


def _profiles_of(config, suffix):
    """committed profile summaries of THIS configuration, oldest first: profiles/<tag>_<config>_<suffix> (tools/prof_summarize.py);
    files of rounds 1-2 carry no configuration tag when they are the channel's"""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_" + suffix))):
        b = os.path.basename(f)
        tagged = [c for c in KNOWN_CONFIGS if f"_{c}_{suffix}" in b]
        if (tagged and tagged[0] == config) or (not tagged and config == "channel"):
            out.append(f)
    return out


def class_traffic(config, ntr=1):
    """HBM bytes per step and kernel class from the newest committed PMC profile of this configuration
    (profiles/*_class_traffic.json, written by tools/prof_summarize.py from separate rocprofv3 --pmc
    passes of this very command); None when there is none."""
    for f in reversed(_profiles_of(config, "class_traffic.json")):
        d = json.load(open(f))
        if d.get("ntr", 1) == ntr:                      # a profile of this very workload
            return d.get("bytes_per_step", {}), os.path.basename(f)
    return None, None


def kernel_traffic(config, name):
    """counted HBM bytes per launch of a kernel in the committed PMC summary of this configuration (2 x FETCH_SIZE + WRITE_SIZE)"""
    for f in reversed(_profiles_of(config, "pmc_hbm_traffic.txt")):
        for line in open(f):
            if line.startswith("#"):
                continue
            parts = line.split()
            if len(parts) >= 5 and " ".join(parts[:-4]).startswith(name):
                return (float(parts[-2]) + float(parts[-1])) * 1e6, os.path.basename(f)
    return None, None


def roofline_of_kernel(kern, kb, F, config):
    """the `roofline` object for the HBM-bound kernel with the longest launch of this run (HIP events around its launches on the
    library's stream): its own algorithmic bytes (kernel_bytes_F) over its average launch duration"""
    cand = {k: v for k, v in kern.items() if k in kb}
    if not cand:
        return None
    k = max(cand, key=lambda q: cand[q][0])
    ms, per_step = cand[k]
    ach = kb[k] * F / (ms * 1e-3) / 1e9
    tr, src = kernel_traffic(config, k)
    r = {"bound": "hbm", "kernel": k, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr,
         "traffic_source": src, "algorithmic_bytes": kb[k] * F, "avg_ms": ms, "launches_per_step": per_step,
         "other_kernels": {q: {"avg_ms": v[0], "frac": kb[q] * F / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS} for q, v in cand.items() if q != k}}
    return r


def dominant_kernel(config):
    """the single largest kernel of the committed rocprofv3 --kernel-trace --stats summary of this configuration
    (profiles/*_kernel_stats.txt of the SAME configuration, newest): name, average us, share of the kernel time"""
    for f in reversed(_profiles_of(config, "kernel_stats.txt")):
        for line in open(f):
            if line.startswith("#") or not line.strip():
                continue
            parts = line.split()
            try:
                return {"name": " ".join(parts[:-4]), "avg_us": float(parts[-3]), "percent_of_kernel_time": float(parts[-1]),
                        "source": os.path.basename(f)}
            except ValueError:
                break
    return None


def usable_cores():
    """Cores this process may actually use: the cgroup CPU quota if there is one (the GPU box shows 256
    logical CPUs but grants 16 cores; 256 threads ran the reference 35x slower than 16), else the CPU count."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, case, masks, nreg, max_seconds=45.0, live=True, full=False, difest=False, start=None):
    """Runs _cpu_baseline in a thread with a 2 GiB stack: the reference keeps its stage-local
    2-D work arrays (21 in remap, ~30 in momtum) on the stack, which at channel size exceeds the
    default 8 MiB limit (BLOM is normally run with `ulimit -s unlimited`)."""
    import threading
    res = {}
    threading.stack_size(2 << 30)
    def body():
        try:
            res.update(_cpu_baseline(cfg, case, masks, nreg, max_seconds, live, full, difest, start))
        except Exception as e:                           # (an exception in a thread would otherwise vanish with its message)
            res["error"] = repr(e)
    th = threading.Thread(target=body)
    th.start()
    th.join()
    threading.stack_size(0)
    return res or {"error": "the reference run did not complete"}


def device_for_bench(case, nreg, masks, live=True, device=0):
    """The device state bench.py times on one tile: config 2's step (full_physics) from the case's initial state with the channel
    experiment's forcing, the diffusivities estimated every step unless live = False (tools/longrun_full_physics.py and the long-run
    parity test start from the same state)."""
    from blom_amd.gpu import BlomGpu
    from blom_amd import hostinit
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks, device=device)
    hostinit.init_state(gpu, case)
    gpu.set("live_slopes", 1)
    hostinit.init_forcing(gpu, case)
    gpu.set("full_physics", 1)
    if live:
        hostinit.init_difest(gpu, case, device=True)
        for d_ in hostinit.DIFEST_NORESM:
            for nm, v in d_.items():
                gpu.set(nm, v)
        gpu.set("difest_live", 1)
    return gpu


def ref_full_init(be, case, xml, xdf):
    """The initial state and the options of the bench workload on a reference backend (the CPU baseline, and
    tools/longrun_reference.py which writes tests/golden/channel_tke_live_long_crc.json from the same state)."""
    from blom_amd import hostinit
    if xdf:
        be.ref.stage("difest_init", *hostinit.step_indices(0, case.kdm))      # (before the state: initke resets the TKE tracers)
    hostinit.init_state(be, case)
    if xml:
        six0 = hostinit.step_indices(0, case.kdm)
        be.ref.stage("mxlayr_init", *six0)
        hostinit.init_forcing(be, case)
        if xdf:
            hostinit.init_difest(be, case)
            for d_ in hostinit.DIFEST_NORESM:
                for nm, v in d_.items():
                    be.ref.set(nm, v)
        for nm, v in dict(rm0=1.2, rm5=0.0, niwgf=0.0, niwbf=0.35, niwlf=0.5, ce=0.06, tau_mlr=86400.0, lfmin=5.0e3, swamxd=200.0, sref=34.65,
                          xmi=0.0, trxday=0.0, srxday=0.0, trxdpt=1.0, srxdpt=1.0, trxlim=1.5, srxlim=0.5).items():
            be.ref.set(nm, float(v))
        be.ref.set("mlrttp", "constant")
        for nm, v in dict(l1mi=11, l2mi=12, l3mi=1, l4mi=2, l5mi=3, aptflx=0, apsflx=0, ditflx=0, disflx=0, srxbal=0, nstep_in_day=96,
                          nday_of_year=1, nday_in_year=365).items():
            be.ref.set(nm, int(v))


# what a step of config 2 reads of the previous one beside the state arrays of tests/parity.py (the reference's restart file carries
# them too, phy/mod_restart.F90): cmnfld's fields, the mixed layer's reservoirs and fluxes, the diffusivity closure's arrays
RESTART_EXTRA = ["bfsqi", "bfsql", "bfsqf", "nslpx", "nslpy", "nnslpx", "nnslpy",
                 "ustar", "ustar3", "idkedt", "uml", "vml", "umlres", "vmlres", "surflx", "sswflx", "surrlx", "salflx", "brnflx", "salrlx",
                 "salt_corr", "trc_corr", "trflx", "mtkeus", "mtkeni", "mtkebf", "mtkers", "mtkepe", "mtkeke", "pbrnda", "buoyfl", "fmltfz",
                 "sfl", "hmltfz", "Prod", "Buoy", "Shear2", "L_scale"]


def continue_from_device(be, gpu, ns, case):
    """Copies the device's state at step count ns into the reference backend `be` so that it continues the run (bench.py --spinup;
    the GPU suite checks that the two then stay bit-identical: tests/test_xcheck_difest.py)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    from parity import copy_state, STATE_FIELDS, INT_FIELDS
    gpu.sync()
    copy_state(gpu, be, fields=[f for f in STATE_FIELDS + INT_FIELDS + RESTART_EXTRA if gpu.has_field(f)])
    be.set("delt1", 2.0 * case.params["baclin"])
    return ns


def _cpu_baseline(cfg, case, masks, nreg, max_seconds=45.0, live=True, full_physics=False, difest=False, start=None):
    """The reference's own Fortran (preferred) or the C restatement, timed on the host for a bounded number of steps.
    Preferred build: oracle/_ref/<cfg>_omp_xed -- the reference's hot-path modules INCLUDING its real mod_cmnfld_routines and
    mod_eddtra (compiled against the two small stand-in modules of oracle/xcheck/, see there), with its OpenMP directives
    on: every stage of the sequence the device is timed on (cmnfld2's slopes from the evolving state, eddtra on them) runs in
    the reference's code on all host cores.  Without it: <cfg>_omp, which lacks those two stages -- they are then timed on the
    C restatement on one thread and reported beside `value`, not inside it."""
    from blom_amd import hostinit
    from blom_amd.stepper import dyncore_step, DYNCORE_STAGES, FULL_STAGES, FULL_STAGES_LIVE
    from oracle.refblom import get_ref_backend, have_ref
    from oracle.coracle import COracle
    ncores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    # --physics full: the build with the reference's real mod_thermf_channel, mod_mxlayr, mod_niw as well (oracle/Makefile *_xml)
    # ... and its real mod_difest (oracle/Makefile *_xdf): the diffusivity estimates live, as on the device
    xdf = full_physics and difest and have_ref(cfg + "_omp_xdf")
    if full_physics and difest and not xdf:
        return {"error": f"oracle/_ref/{cfg}_omp_xdf/libblomref.so is missing: no CPU baseline for the step with live diffusivities"}
    xml = xdf or (full_physics and have_ref(cfg + "_omp_xml"))
    if full_physics and not xml:
        return {"error": f"oracle/_ref/{cfg}_omp_xml/libblomref.so is missing: no CPU baseline for --physics full"}
    full = xml or have_ref(cfg + "_omp_xed")
    ref_cfg = cfg + "_omp_xdf" if xdf else cfg + "_omp_xml" if xml else cfg + "_omp_xed" if full else (cfg + "_omp" if have_ref(cfg + "_omp") else (cfg if have_ref(cfg) else None))
    stages = DYNCORE_STAGES
    note, de, dc = "", 0.0, 0.0
    if ref_cfg is not None and not full:
        # (The restatement's j-loops carry OpenMP pragmas, but on the GPU box gcc's runtime next to PyTorch's and LLVM's thread
        # pools ran them 2-3x SLOWER on 16 threads than on one, so they are timed on one thread, before the reference's
        # library and its thread team exist.)
        import ctypes
        try:
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
        except OSError:
            pass
        co = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
        hostinit.init_state(co, case)
        six = hostinit.step_indices(1, case.kdm)
        co.set("delt1", 2 * case.params["baclin"])

        def best_of(stage, reps=4):              # the fastest of a few repetitions: the host cores are shared
            best = 1e30
            for _ in range(reps):
                t1 = time.perf_counter()
                co.stage(stage, *six)
                best = min(best, time.perf_counter() - t1)
            return best
        co.stage("eddtra", *six)
        de = best_of("eddtra")
        co.stage("cmnfld2", *six)
        dc = best_of("cmnfld2")
        del co
        note = (f"; not in value: eddtra ({de * 1e3:.1f} ms) and cmnfld2's slopes ({dc * 1e3:.1f} ms), which this reference build "
                "lacks, timed on the C restatements on one thread")
    kind, cores = None, 1
    try:
        if ref_cfg is not None:
            if "_omp" in ref_cfg:               # the reference with its OpenMP directives on, all host cores
                cores = ncores
                os.environ.setdefault("OMP_PROC_BIND", "close")
                os.environ["OMP_STACKSIZE"] = "1G"   # the stages keep private 2-D work arrays on the thread stacks
            be = get_ref_backend(ref_cfg, case.depth, ntr=case.ntr)      # (the reference's tracer count is a run-time quantity)
            kind = "reference"
            if full:
                be.ref.set("eitmth", "gm")
                be.has_stage = lambda name: True             # this build's harness knows eddtra and cmnfld2
                if xml:
                    stages = FULL_STAGES_LIVE if xdf else FULL_STAGES
                elif live:
                    stages = tuple("cmnfld2" if s_ == "halo_cmnfld2" else s_ for s_ in DYNCORE_STAGES)
                note = ("; every stage in the reference's own code, mod_cmnfld_routines and mod_eddtra compiled against the "
                        "stand-in modules of oracle/xcheck/ (one array of mod_difest, the diagnostic flags of mod_dia)")
    except Exception:
        kind, cores, de, dc, note, full, stages = None, 1, 0.0, 0.0, "", False, DYNCORE_STAGES
    if kind is None:
        be = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
        kind = "port"
    ref_full_init(be, case, xml, xdf)
    if start is not None:
        # --spinup: the reference continues from the state the device holds (every array of the step, both time levels)
        ns = continue_from_device(be, *start, case)
    else:
        ns = dyncore_step(be, 0, case.params["baclin"], stages=stages)          # forward first step (untimed)
    # per-stage host times beside the device's stages_ms: the hook fires before every stage
    per_stage, mark = {}, [None, 0.0]

    def hook(st, six):
        now = time.perf_counter()
        if mark[0] is not None:
            per_stage[mark[0]] = per_stage.get(mark[0], 0.0) + (now - mark[1])
        mark[0], mark[1] = st, now
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < max_seconds and n < 120):
        ns = dyncore_step(be, ns, case.params["baclin"], hook=hook, stages=stages)
        hook(None, None)
        n += 1
    dt = (time.time() - t0) / n
    ref_only_ms = dt * 1e3
    how = (f"{cores} OpenMP threads (reference built with -fopenmp)" if cores > 1 else
           "single thread (reference built without OpenMP)" if kind == "reference" else "single thread (C restatement)")
    stages_ms = {k: round(v / n * 1e3, 2) for k, v in per_stage.items() if k}
    # the device reports advect as "remap" (or "cppm"): same name here
    if "advect" in stages_ms:
        stages_ms[case.params.get("advmth", "remap")] = stages_ms.pop("advect")
    if "cmnfld2" in stages_ms:
        stages_ms["cmnfld"] = stages_ms.pop("cmnfld2") + stages_ms.pop("cmnfld1", 0.0)
    if "difest_isobml_pre" in stages_ms:
        stages_ms["difest"] = stages_ms.pop("difest_isobml_pre")
    return dict(value=case.params["baclin"] / 86400.0 / dt, unit="simulated-days/sec", cores=cores, kind=kind, stages_ms=stages_ms,
                reference_only_ms=round(ref_only_ms, 2), steps_timed=n,
                build=ref_cfg,
                restatement_legs=(None if full or kind != "reference" else
                                  {"eddtra_ms": round(de * 1e3, 2), "cmnfld2_ms": round(dc * 1e3, 2), "threads": 1, "in_value": False}),
                sample=f"{n} baroclinic steps of the same {cfg} workload, {dt * 1e3:.1f} ms/step, {how}{note}")


def bench_hor3map(args):
    """`--config hor3map`: SURVEY.md 8 row a5 on its own.  A step is the six-call PPM sequence of one tracer-type field
    (prepare_reconstruction, reconstruct, extract_polycoeff, regrid, prepare_remapping, remap -- what
    phy/mod_ale_regrid_remap.F90:224-247, :405-415, :1038-1046 does per column) on a model-like slab of the channel's
    106 080 columns x 53 layers, inputs resident in HBM.  One JSON line; roofline: the slowest of the six kernels against
    its caller-visible bytes."""
    import numpy as np
    import torch
    import h3m_cases as hc
    from blom_amd import hor3map as h3
    ncol, n = 106080, 53
    x, u, xd, ug = hc.make_slab(11, ncol, n, n, n + 1)
    dev = torch.device("cuda:0")
    tx, tu, txd, tug = (torch.from_numpy(a).to(dev) for a in (x, u, xd, ug))
    cfg = (hc.PPM, 6, 4, hc.NON_OSCILLATORY_POSDEF, True, False)
    g = h3.ReconGrid(ncol, n, cfg[0], cfg[1], cfg[2])
    g.set_io(device_pointers=True, check_errors=False)
    s_, r = h3.ReconSrc(g, cfg[3], cfg[4], cfg[5]), h3.Remap(g, n)
    npc = h3.P_ORD[cfg[0]] + 1
    tpc = torch.empty((ncol, n, npc), dtype=torch.float64, device=dev)
    tud = torch.empty((ncol, n), dtype=torch.float64, device=dev)
    txg = torch.empty((ncol, n + 1), dtype=torch.float64, device=dev)
    calls = [("prepare_reconstruction", lambda: g.prepare_reconstruction(tx.data_ptr()), 1),
             ("reconstruct", lambda: s_.reconstruct(tu.data_ptr()), 1),
             ("extract_polycoeff", lambda: s_.extract_polycoeff(out=tpc.data_ptr()), npc),
             ("regrid", lambda: s_.regrid(tug.data_ptr(), -1e33, h3.REGRID_METHOD_1, out=txg.data_ptr(), n_grd=n + 1), 2),
             ("prepare_remapping", lambda: r.prepare_remapping(txd.data_ptr()), 1),
             ("remap", lambda: r.remap(s_, out=tud.data_ptr()), 1)]
    F = ncol * n * 8.0
    for _ in range(max(1, args.warmup)):
        for _, f, _ in calls:
            f()
    g.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _, f, _ in calls:
            f()
    g.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = {}
    for nm, f, _ in calls:                     # per-kernel HIP-event times (the library's own events), median of 9
        v = []
        for _ in range(9):
            f()
            g.sync()
            v.append(g.last_kernel_ms())
        kms[nm] = float(np.median(v))
    dom = max(kms, key=kms.get)
    alg = {nm: nf * F for nm, _, nf in calls}
    out = {"metric": "hor3map PPM six-call sequences of a 106080 x 53 slab per second", "value": args.steps / dt,
           "unit": "slab-sequences/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "mod_hor3map API (phy/mod_hor3map.F90:3834-4559), PPM / non-oscillatory posdef limiting / boundary "
                                  "orders 6, 4: prepare_reconstruction, reconstruct, extract_polycoeff, regrid (method 1, the default BLOM's ale_regrid_remap calls), "
                                  "prepare_remapping, remap on 106080 columns x 53 layers (tests/h3m_cases.py: make_slab)"},
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": alg[dom] / (kms[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": alg[dom] / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes": alg[dom], "avg_ms": kms[dom],
                        "note": "caller-visible arrays in + out; the kernels also move their work planes (DESIGN.md 3g: counted "
                                "0.4-0.9 GB per kernel before the store cuts) -- with every walk's loads issued ahead they run at "
                                "4-5 TB/s of counted traffic"},
           "kernels_ms": {k: round(v, 4) for k, v in kms.items()}}
    if not args.no_cpu_baseline and hc.have_ref():
        m = 6000
        t0 = time.perf_counter()
        hc.run_ref(*cfg, x[:m].copy(), u[:m].copy(), xd[:m].copy(), ug[:m].copy(), hc.METHOD_1)
        dtc = (time.perf_counter() - t0) / m * ncol
        out["cpu_baseline"] = {"value": 1.0 / dtc, "unit": "slab-sequences/sec", "cores": 1, "kind": "reference",
                               "sample": f"the reference's compiled mod_hor3map on {m} of the slab's columns, one core, scaled to the slab"}
    g.free()
    print(json.dumps(out))


def bench_ale(args):
    """`--config ale`: SURVEY.md 8 row f3's first piece on its own -- one ale_regrid_remap (phy/mod_ale_regrid_remap.F90:1486) of
    BASELINE's channel, vcoord_type = 'cntiso_hybrid' with regrid_method = 'direct', the options of the reference's
    tests/fuk95/limits: reconstruct T, S and the tracers, regrid the interfaces to their target densities, remap T, S, tracers,
    u and v.  State: the isopycnic state after two steps of the dynamical core.  One JSON line; the CPU baseline is the
    reference's real module (oracle/_ref/channel_tke_omp_xale, cross-check build) when it has been built."""
    import numpy as np
    import torch
    from blom_amd.gpu import BlomGpu
    from blom_amd import hostinit
    case, nreg, masks = build_case("channel")
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(gpu, case)
    ns = gpu.step(0, 2)
    kk = case.kdm
    pbot = float(np.max(gpu.get("p")[kk][masks["ip"] > 0]))
    gpu.set("vcoord_type", "cntiso_hybrid")
    gpu.set("ale_regrid_method", "direct")
    gpu.set_vector("plevel", 0.05 * pbot * (np.arange(kk) / kk) ** 1.3)
    six = hostinit.step_indices(ns, kk)
    for _ in range(max(1, args.warmup)):
        gpu.stage("ale_regrid_remap", *six)
    gpu.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gpu.stage("ale_regrid_remap", *six)
    gpu.sync()
    dt = (time.perf_counter() - t0) / args.steps
    F = case.idm * case.jdm * kk * 8.0
    ntl = 2 + case.ntr
    alg = (ntl + 1 + 2 + 2) * F + (ntl + 2 + 2 + 2 + 2) * F     # read dp, T, S, trc, u, v, dpu, dpv; write those, sigma, dpuold, dpvold
    out = {"metric": "ale_regrid_remap calls per second", "value": 1.0 / dt, "unit": "calls/sec", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"ale_regrid_remap on channel {case.idm}x{case.jdm}x{kk}, ntr={case.ntr}: vcoord_type = cntiso_hybrid, "
                                  "regrid_method = direct, ppm / non_oscillatory, boundary orders 6 / 4 (the reference's tests/fuk95/limits); "
                                  "every hor3map call synchronises for its error status",
                      "parity": "cross-checked against the reference's real module built against a stand-in for mod_dia (tests/test_xcheck_ale.py)"},
           "roofline": {"bound": "hbm", "kernel": "ale_regrid_remap", "achieved": alg / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": alg, "avg_ms": dt * 1e3,
                        "note": "the model's fields in and out once; the column kernels of the engine are latency bound (DESIGN.md 3a)"}}
    gpu.close()
    print(json.dumps(out))


HYB_OPTS = dict(reconstruction_method="ppm", upper_bndr_ord=6, lower_bndr_ord=4, tracer_limiting="non_oscillatory",
                velocity_limiting="non_oscillatory", tracer_pc_upper_bndr=True, tracer_pc_lower_bndr=False,
                velocity_pc_upper_bndr=True, velocity_pc_lower_bndr=False, regrid_method="nudge")      # the group of tests/fuk95/limits


def _hybrid_fields(case):
    """what the parts of the hybrid step that are not built would produce, constant in time: vertical diffusivities, non-local
    fractions, shortwave absorption, surface fluxes, boundary layer depth"""
    import numpy as np
    kk, nj, ni = case.kdm, case.jdm + 8, case.idm + 8
    z = np.arange(kk + 1)[:, None, None] / kk
    frac = np.clip(1.0 - z / 0.2, 0.0, 1.0) ** 2 * np.ones((1, nj, ni))
    f = {}
    for nm in ("kvisc_m", "kdiff_t", "kdiff_s"):
        f[nm] = 1e-5 + 1e-3 * np.exp(-((z - 0.03) / 0.05) ** 2) * np.ones((1, nj, ni))
    for nm in ("t_ns_nonloc", "s_nb_nonloc", "t_rs_nonloc", "s_rs_nonloc", "mu_nonloc", "mv_nonloc"):
        f[nm] = frac
    for nm, v in (("swfc1", .6), ("swfc2", .4), ("swal1", 1.), ("swal2", 15.), ("surflx", -40.), ("sswflx", -60.), ("salflx", 5e-4),
                  ("OBLdepth", 40.), ("surrlx", 0.), ("brnflx", 0.), ("salrlx", 0.), ("salt_corr", 0.)):
        f[nm] = v * np.ones((1, nj, ni))
    if case.ntr:                         # (the reference's arrays start from its initialisation patterns, not from zero)
        f["trflx"] = np.zeros((case.ntr, nj, ni))
        f["trc_corr"] = np.zeros((case.ntr, nj, ni))
    return f


def _cpu_baseline_hybrid(case, masks, nreg, plevel, max_seconds=40.0, neutral=True):
    """the same hybrid stage sequence in the reference's own modules (oracle/_ref/channel_tke_omp_xaln: real mod_ale_regrid_remap,
    mod_ale_forcing, mod_ale_vdiff, mod_eddtra, mod_cmnfld_routines behind the stand-ins of oracle/xcheck), OpenMP on all cores"""
    import ctypes as C
    import tempfile
    import numpy as np
    from blom_amd import hostinit
    from blom_amd.stepper import dyncore_step, HYBRID_STAGES
    from oracle.refblom import get_ref_backend, have_ref
    lib = "channel_tke_omp_xaln"
    if not have_ref(lib):
        return {"error": f"oracle/_ref/{lib}/libblomref.so is missing"}
    ncores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ["OMP_STACKSIZE"] = "1G"
    be = get_ref_backend(lib, case.depth)
    be.has_stage = lambda name: True
    hostinit.init_state(be, case)
    for nm, a in _hybrid_fields(case).items():
        be.put(nm, a)
    kk = case.kdm
    ierr = C.c_int(0)
    v = np.ascontiguousarray(plevel, dtype=np.float64)
    be.ref.lib.ref_set_vec(b"plevel", v.ctypes.data_as(C.c_void_p), C.c_int(kk), C.byref(ierr))
    be.ref.set("vcoord_tag", 2)
    be.ref.set("swamxd", 200.0)
    six0 = hostinit.step_indices(0, kk)
    o = HYB_OPTS
    fl = lambda b_: ".true." if b_ else ".false."
    txt = (" &ALE_REGRID_REMAP\n"
           f"  RECONSTRUCTION_METHOD  = '{o['reconstruction_method']}'\n  UPPER_BNDR_ORD = {o['upper_bndr_ord']}\n"
           f"  LOWER_BNDR_ORD = {o['lower_bndr_ord']}\n  DENSITY_LIMITING = 'monotonic'\n"
           f"  TRACER_LIMITING = '{o['tracer_limiting']}'\n  VELOCITY_LIMITING = '{o['velocity_limiting']}'\n"
           f"  TRACER_PC_UPPER_BNDR = {fl(o['tracer_pc_upper_bndr'])}\n  TRACER_PC_LOWER_BNDR = {fl(o['tracer_pc_lower_bndr'])}\n"
           f"  VELOCITY_PC_UPPER_BNDR = {fl(o['velocity_pc_upper_bndr'])}\n  VELOCITY_PC_LOWER_BNDR = {fl(o['velocity_pc_lower_bndr'])}\n"
           f"  REGRID_METHOD = '{o['regrid_method']}'\n /\n")
    be.ref.set("ltedtp_opt", 2)                 # the structures' index range (phy/mod_ale_regrid_remap.F90:1384-1390)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "limits"), "w").write(txt)
        cwd = os.getcwd()
        os.chdir(td)
        try:
            be.ref.stage("ale_init", *six0)
        finally:
            os.chdir(cwd)
    be.ref.stage("eddtra_init_fox08", *six0)
    be.ref.set("eitmth", "gm")
    be.ref.set("ltedtp_opt", 2 if neutral else 1)
    be.ref.set("ndiff_surface_align", 1)
    be.ref.stage("cmnfld1", *hostinit.init_indices(0, kk))
    ns = dyncore_step(be, 0, case.params["baclin"], stages=HYBRID_STAGES)
    per_stage, mark = {}, [None, 0.0]

    def hook(st, six):
        now = time.perf_counter()
        if mark[0] is not None:
            per_stage[mark[0]] = per_stage.get(mark[0], 0.0) + (now - mark[1])
        mark[0], mark[1] = st, now
    t0, n = time.time(), 0
    while n < 2 or (time.time() - t0 < max_seconds and n < 40):
        ns = dyncore_step(be, ns, case.params["baclin"], hook=hook, stages=HYBRID_STAGES)
        hook(None, None)
        n += 1
    dt = (time.time() - t0) / n
    be.ref.set("vcoord_tag", 1)
    be.ref.set("ltedtp_opt", 1)
    tt = be.get("temp")[:, 4:-4, 4:-4]
    tt = tt[np.broadcast_to((masks["ip"][4:-4, 4:-4] > 0)[None], tt.shape)]
    if not (np.isfinite(tt).all() and np.abs(tt).max() < 100.0):
        return {"error": "the reference's state left the physical range during the timed steps"}
    return dict(value=case.params["baclin"] / 86400.0 / dt, unit="simulated-days/sec", cores=ncores, kind="reference", build=lib,
                stages_ms={k: round(v_ / n * 1e3, 2) for k, v_ in per_stage.items() if k}, steps_timed=n,
                sample=f"{n} steps of the same hybrid stage sequence on the channel, {dt * 1e3:.1f} ms/step, {ncores} OpenMP threads; every stage in "
                       "the reference's own code, the modules that import netCDF-bound ones compiled against the stand-ins of oracle/xcheck "
                       "(cppm's sweeps, which the reference does not thread, are not in this sequence: advmth = remap)")


def bench_hybrid_step(args):
    """`--config hybrid`: the step of the hybrid vertical coordinate as far as it is built on
    BASELINE's channel: ale_regrid_remap (cntiso_hybrid, nudge, ppm: the options of the reference's tests/fuk95/limits), cmnfld2,
    eddtra (eddtra_ale: Gent-McWilliams + fox08), advect (remap), pbcor1, diffus, pgforc, momtum, cmnfld_bfsqi_ale, ale_forcing, ale_vdifft,
    ale_vdiffm, barotp, pbcor2, tmsmt2, cmnfld1; vertical diffusivities, non-local fractions, boundary layer depth and surface fluxes
    constant in time (their producers need CVMix / forcing files).  One JSON line with `roofline` (the stage class with the largest
    time against its algorithmic bytes) and `cpu_baseline` (the same stages in the reference's modules, channel_tke_omp_xaln)."""
    import numpy as np
    import threading
    from blom_amd.gpu import BlomGpu
    from blom_amd import hostinit
    case, nreg, masks = build_case("channel")
    gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
    hostinit.init_state(gpu, case)
    kk = case.kdm
    for nm, a in _hybrid_fields(case).items():
        gpu.put(nm, a)
    pbot = float(np.max(gpu.get("p")[kk][4:-4, 4:-4][masks["ip"][4:-4, 4:-4] > 0]))
    plevel = 0.05 * pbot * (np.arange(kk) / kk) ** 1.3
    gpu.set("vcoord_type", "cntiso_hybrid")
    o = HYB_OPTS
    gpu.set("ale_regrid_method", o["regrid_method"])
    gpu.set("ale_reconstruction_method", o["reconstruction_method"])
    gpu.set("ale_tracer_limiting", o["tracer_limiting"])
    gpu.set("ale_velocity_limiting", o["velocity_limiting"])
    for nm in ("upper_bndr_ord", "lower_bndr_ord"):
        gpu.set("ale_" + nm, int(o[nm]))
    for nm in ("tracer_pc_upper_bndr", "tracer_pc_lower_bndr", "velocity_pc_upper_bndr", "velocity_pc_lower_bndr"):
        gpu.set("ale_" + nm, 1 if o[nm] else 0)
    gpu.set("mlrmth", "fox08")
    gpu.set_vector("plevel", plevel)
    neutral = args.ltedtp == "neutral"
    gpu.set("ltedtp_opt", 2 if neutral else 1)
    gpu.set("ndiff_surface_align", 1)
    for o_ in args.opt:
        nm, v = o_.split("=")
        gpu.set(nm, int(v))
    gpu.stage("cmnfld1", *hostinit.init_indices(0, kk))
    ns = gpu.step(0, max(4, args.warmup))
    gpu.sync()
    prof_file = os.environ.get("BLOM_NDIFF_PROF")          # debug: per-wave timestamps of k_ndiff_flux (tools/ndiff_waves.py reads them)
    if prof_file:
        import ctypes as C
        nw = 6 * 2 * ((gpu.get("p").shape[1] * gpu.get("p").shape[2] + 63) // 64)
        gpu.lib.blomgpu_dbg_kprof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        assert gpu.lib.blomgpu_dbg_kprof(gpu.ctx, None, nw) == 0
        gpu.set("use_graph", 0)                    # (setting an option drops any captured graphs: their launches carry the old buffer pointer)
    t0 = time.perf_counter()
    ns = gpu.step(ns, args.steps)
    gpu.sync()
    if prof_file:
        buf = (C.c_longlong * nw)()
        assert gpu.lib.blomgpu_dbg_kprof(gpu.ctx, buf, nw) == 0
        np.save(prof_file, np.frombuffer(buf, dtype=np.int64).reshape(-1, 6).copy())
    dt = (time.perf_counter() - t0) / args.steps
    # per-class HIP-event times over a few more steps
    gpu.set("timing", 1)
    gpu.timer_reset()
    ns = gpu.step(ns, min(args.steps, 5))
    gpu.sync()
    classes = ["ale_regrid_remap", "ndiff", "cmnfld", "eddtra", "remap", "diffus", "pgforc", "momtum", "ale_forcing", "ale_vdiff", "barotp", "pbcor1", "pbcor2"]
    live = {}
    for cl in classes:
        ms, n = gpu.timer_get(cl)
        if n:
            live[cl] = ms / min(args.steps, 5)           # ms per step of the class (ale_vdiff: its two stages)
    gpu_timers = {kn: gpu.timer_get(kn) for kn in ("k_ndiff_prep", "k_ndiff_flux", "k_ndiff_eval", "k_ndiff_uvflx", "k_ndiff_apply")}
    gpu.set("timing", 0)
    u = gpu.get("u")[:, 4:-4, 4:-4]
    finite = bool(np.isfinite(u[np.broadcast_to((masks["iu"][4:-4, 4:-4] > 0)[None], u.shape)]).all())
    baclin = case.params["baclin"]
    ntr = case.ntr
    F = case.idm * case.jdm * kk * 8.0
    # algorithmic bytes per class in units of F (every distinct 3-D array read or written once): the classes of the isopycnic step as
    # in SURVEY.md 8(d) (momtum + 2 for mu_nonloc, mv_nonloc), and
    #   ale_regrid_remap  R dp, T, S, sigma, tracers, u, v (n); W dp, T, S, sigma, tracers, u, v, dpu, dpv, dpuold, dpvold, p, pu, pv
    #   ale_vdiff         R dp, T, S, tracers, u, v, dpu, dpv, Kdiff_t, Kdiff_s, Kvisc_m, the six non-local fractions; W T, S, tracers, sigma, u, v
    #   ale_forcing       R dp, T, S, p; W t_sw_nonloc, s_br_nonloc, buoyfl
    cb = dict(class_bytes_F(ntr, ntr_diffused(case)))
    cb["momtum"] += 2
    cb.update(ale_regrid_remap=(6 + ntr) + (13 + ntr), ale_vdiff=(16 + ntr) + (5 + ntr), ale_forcing=4 + 3)
    if neutral:
        #   ndiff (inside ale_regrid_remap)  R source and destination interfaces (2), the reconstruction coefficients of T, S, tracers
        #                     (3 per layer with ppm), T, S, tracers (n), difiso, pu, pv; W the flux convergence per field, utflld .. vsflld (4),
        #                     nslpx, nslpy; RW utflx .. vsflx (4)
        nloc = ntr + 2
        cb["ndiff"] = 2 + 3 * nloc + nloc + 1 + 2 + nloc + 4 + 2 + 2 * 4
        cb["ale_regrid_remap"] += cb["ndiff"]
    nd_k = {}
    for kn in ("k_ndiff_prep", "k_ndiff_flux", "k_ndiff_eval", "k_ndiff_uvflx", "k_ndiff_apply"):
        ms, n = gpu_timers.get(kn, (0.0, 0))
        if n:
            nd_k[kn] = ms / n
    hbm = {k: v for k, v in live.items() if k in cb and k != "ndiff"}        # (ndiff's time lies inside ale_regrid_remap's)
    dom = max(hbm, key=hbm.get)
    ach = cb[dom] * F / (live[dom] * 1e-3) / 1e9
    tot = sum(cb[k] for k in hbm) * F
    out = {"metric": "simulated-days/sec", "value": baclin / 86400.0 / dt, "unit": "simulated-days/sec", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"channel {case.idm}x{case.jdm}x{kk}, ntr={case.ntr}: the step of vcoord_type = cntiso_hybrid as far as built "
                                  f"(ltedtp = '{args.ltedtp}'; ale_regrid_remap nudge/ppm 6/4" + (" with neutral diffusion" if neutral else "") + ", cmnfld2, eddtra_ale gm+fox08, advect remap, pbcor1, diffus, pgforc, momtum, cmnfld_bfsqi_ale, ale_forcing, "
                                  "ale_vdifft, ale_vdiffm, barotp, pbcor2, tmsmt2, cmnfld1); diffusivities, non-local fractions, boundary layer depth, surface fluxes constant",
                      "parity": "cross-checked stage sequence, also at this size (tests/test_xcheck_hybrid_step.py)", "state_finite": finite},
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": None, "algorithmic_bytes": cb[dom] * F, "avg_ms": live[dom],
                        "note": "stage class (its kernels are launched back to back); HIP events on the library's stream" +
                                ("; with ltedtp = 'neutral' the class contains the neutral diffusion, whose searches are bound by instruction issue and "
                                 "load latency, not by bytes (DESIGN.md 3j): its kernels run on a stream of their own beside the remapping" if neutral else ""),
                        "step_hbm_frac": tot / dt / 1e9 / HBM_PEAK_GBS},
           "stages_ms": live}
    if nd_k:
        out["ndiff_kernels_ms"] = nd_k
    gpu.close()
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if not args.no_cpu_baseline:
        res = {}
        threading.stack_size(2 << 30)
        th = threading.Thread(target=lambda: res.update(_cpu_baseline_hybrid(case, masks, nreg, plevel, neutral=neutral)))
        th.start()
        th.join()
        threading.stack_size(0)
        out["cpu_baseline"] = res or {"error": "the reference run did not complete"}
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="channel")
    ap.add_argument("--advmth", default="remap", choices=["remap", "cppm"],
                    help="advection method (the reference's advmth); the headline configuration is remap")
    ap.add_argument("--tracers", default="default",
                    help="default: the reference's default option set (TKE, its advection, ideal age: ntr = 3); "
                         "iage: ideal age only (ntr = 1); an integer N > 3: the default set plus N - 3 passive tracers "
                         "(BASELINE.json config 5 advects iHAMOCC's tracers through the same stages)")
    ap.add_argument("--slopes", default="live", choices=["live", "frozen"],
                    help="live: cmnfld2 computes the neutral slopes eddtra consumes every step (phy/mod_cmnfld_routines.F90:1158); "
                         "frozen: round 1's analytic pattern of amplitude NSLP0")
    ap.add_argument("--physics", default="full", choices=["full", "dyncore"],
                    help="full (default): config 2's sequence as far as built -- cmnfld2, the built part of difest_isobml "
                         "(ustar3, niw_ke_tendency), thermf, mxlayr, cmnfld1 besides the dynamical core (stepper.FULL_STAGES); on tiles it "
                         "needs --barotp replicated (thermf's global sums are formed on that solve's global context); "
                         "dyncore: the dynamical core alone, as in rounds 1-3 (also what --barotp decomposed and the weak-scaling layout run)")
    ap.add_argument("--ltedtp", default="neutral", choices=["neutral", "layer"],
                    help="--config hybrid: lateral tracer diffusion, 'neutral' (phy/mod_ndiff.F90 inside ale_regrid_remap; the reference's "
                         "default for cntiso_hybrid) or 'layer' (diffus)")
    ap.add_argument("--frozen-diffusivities", action="store_true",
                    help="--physics full: leave out the diffusivity estimates of difest_isobml (difint, difiso, difdia, difwgt stay at their "
                         "initial values, as in round 4)")
    ap.add_argument("--forcing", default="default", choices=sorted(FORCING),
                    help="default: the channel experiment's own forcing (channel/mod_channel.F90:365: ustarw = 0.005, which "
                         "thermf_channel's factor 1e2 turns into a friction velocity of 0.5 m/s: the mixed layer reaches 1.6 km within 200 steps); "
                         "calm: this project's second published set, ustarw = 5e-5 (0.005 m/s after the factor).  The headline uses `default`")
    ap.add_argument("--spinup", type=int, default=0,
                    help="steps integrated before the warm-up (default 0: the timed window starts from rest, steps W+1..W+K); with N > 0 the "
                         "CPU baseline starts from the state the device holds at the end of the run, downloaded")
    ap.add_argument("--blocks", type=int, default=5,
                    help="the timed region is block 1 (the headline: `value`, `ms_per_step`); blocks 2..B of --steps steps each follow and "
                         "ms_per_step_median / _min / _max over all B blocks are reported beside it")
    ap.add_argument("--spunup-steps", type=int, default=1000,
                    help="one tile, --physics full: after the measurement the run is continued to this step count and --steps steps are "
                         "timed again (key `spunup`: a state a model run is in, not the transient from rest); 0: leave it out")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dyncore-compare", action="store_true",
                    help="leave out the extra steps of the dynamical core alone that follow the measurement (profiling runs)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                    help="library option for A/B runs of kernel variants, e.g. barotp_tile=3216 (default: production kernels)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="N > 1: strong (default) = the BASELINE domain cut into tiles; weak = N times as long a channel")
    ap.add_argument("--tiles", default=None, metavar="NPXxNPY", help="tile grid of the strong-scaling run (default: by N)")
    ap.add_argument("--barotp", default="replicated", choices=["replicated", "decomposed"],
                    help="tiles (N > 1 or --tiles): replicated = every rank gathers barotp's 2-D inputs once per step and solves "
                         "the whole barotropic domain itself; decomposed = the reference's scheme, one exchange per substep pair")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 run: nccl (= RCCL, one GPU per rank) or gloo -- the CPU rehearsal of the "
                         "same launcher / rendezvous / tile code with BLOMGPU_LIB pointing at the host emulation of the library "
                         "(tests/hostemu, TEST use: tests/test_multirank_host.py); no torch.cuda call is made then")
    ap.add_argument("--rccl-self", action="store_true",
                    help="N=1 only: route the halo update through the RCCL transport (rank sends to itself) "
                         "to measure the exchange overhead of the N>1 path on one GPU")
    args = ap.parse_args()
    if args.config == "hor3map":
        return bench_hor3map(args)
    if args.config == "ale":
        return bench_ale(args)
    if args.config == "hybrid":
        return bench_hybrid_step(args)

    from blom_amd import launch
    env = launch.rank_env()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as `python bench.py --gpus N`: start the N ranks (one process per GPU) as children of this process,
        # which has not touched the GPU, and pass rank 0's JSON line on
        import subprocess
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus != env.world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {env.world}")
    rank, world, local = env.rank, env.world, env.local
    scaling = (args.scaling or "strong") if (world > 1 or args.tiles) else "weak"
    import torch
    on_gpu = args.backend == "nccl"
    if not on_gpu:
        local = 0                                  # the emulated device
    cuda_sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if world > 1:
        import torch.distributed as dist
        if on_gpu:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    from blom_amd.gpu import BlomGpu, rccl_unique_id
    from blom_amd import hostinit
    case, nreg, masks = build_case(args.config, args.advmth, args.tracers, args.forcing)
    layout = None
    difest_live = False
    full_req = args.physics == "full" and not args.rccl_self and args.slopes == "live"
    tile_area = None
    if (world > 1 and scaling == "strong") or args.tiles:       # --tiles 1x1 at N = 1: the same code path with one rank
        # BASELINE.json configs 3/4: the same domain, npx x npy tiles (bld/blom_dimensions:104-148), one per GPU.
        from blom_amd.tiles import TileLayout, scatter_to_tile
        default_grid = {"channel": {2: (1, 2), 4: (2, 2), 8: (2, 4)}, "tnx2v1s": {2: (2, 1), 4: (2, 2), 8: (4, 2)},
                        "tnx1v4s": {2: (2, 1), 4: (2, 2), 8: (4, 2)}}
        if args.tiles:
            npx, npy = (int(x) for x in args.tiles.lower().split("x"))
        else:
            npx, npy = default_grid.get(args.config, {}).get(world, (1, world))
        if npx * npy != world:
            raise SystemExit(f"bench.py: {npx}x{npy} tiles for {world} GPUs")
        layout = TileLayout.regular(case.idm, case.jdm, npx, npy)
        px, py = layout.rank_tile(rank)
        i0, j0, tii, tjj = layout.tile(px, py)
        # Rank 0 initialises the whole domain on its GPU (the initialisation runs stages on the device) and hands every field to the
        # ranks (launch.FieldsFromRank0: one broadcast per field, GPU to GPU with nccl), each of which keeps its window: the other
        # ranks never hold more than their tile.  (Until round 5 every rank built the whole domain for a moment.)
        whole = None
        if rank == 0:
            whole = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks, device=local)
            hostinit.init_state(whole, case)
            if full_req and args.barotp == "replicated":
                # config 2's step on tiles (config 3): thermf's global sums are formed on the replicated solve's global context
                hostinit.init_forcing(whole, case)
                tile_area = hostinit.ocean_area(whole, case)
                if not args.frozen_diffusivities:
                    hostinit.init_difest(whole, case, device=True)         # (the planes travel to the tile with the other fields)
        if full_req and args.barotp == "replicated":
            tile_area = launch.broadcast_object(tile_area, env)
        src = launch.FieldsFromRank0(whole, env, device=f"cuda:{local}" if on_gpu else "cpu")
        gpu = BlomGpu(tii, tjj, case.kdm, case.ntr, nreg, {k: layout.window(masks[k], px, py) for k in masks}, device=local,
                      itdm=case.idm, jtdm=case.jdm, i0=i0, j0=j0)
        for nm, v in case.params.items():
            if not nm.endswith("0"):
                gpu.set(nm, v)
        scatter_to_tile(src, gpu, layout, px, py)
        # every rank solves the whole 2-D barotropic domain on a second context (no exchange inside barotp's substep loop)
        glob = None
        if args.barotp == "replicated":
            from blom_amd.tiles import make_barotp_global
            glob = make_barotp_global(src, case, masks, device=local)
        if whole is not None:
            whole.close()
        gpu.set("delt1", case.params["baclin"])
        gpu.rccl_init_2d(launch.share_unique_id(rccl_unique_id, env), rank, npx, npy)
        if glob is not None:
            gpu.rccl_attach_barotp_global(glob, layout.isizes, layout.jsizes)
    elif world > 1:
        # Weak scaling: the channel is made `world` times as long in i (its bathymetry repeated
        # with the tile's period) and cut into `world` tiles along i, one per GPU.  Every tile
        # then starts from the single-tile state -- its periodic wrap IS its neighbours' data --
        # and the halos travel between GPUs over RCCL (blom_amd/csrc/comm_rccl.hip).
        if nreg in (0, 4):
            raise SystemExit("bench.py --gpus N>1 needs a configuration that is periodic in i")
        lay = launch.tile_layout(case.idm, rank, world)
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks, device=local,
                      itdm=lay["itdm"], jtdm=case.jdm, i0=lay["i0"], j0=0)
        gpu.rccl_init(launch.share_unique_id(rccl_unique_id, env), rank, world)
    else:
        gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks, device=local)
        if args.rccl_self:
            gpu.rccl_init(rccl_unique_id(), 0, 1)
    if layout is None:
        hostinit.init_state(gpu, case)
    gpu.set("live_slopes", 1 if args.slopes == "live" else 0)
    # config 2's step as far as built: on one tile, and on tiles that carry the replicated barotropic solve's global context
    # (thermf's sums); the weak-scaling layout and --barotp decomposed run the dynamical-core sequence
    full = full_req and ((world == 1 and layout is None) or tile_area is not None)
    if full:
        # the channel experiment's own forcing (channel/mod_channel.F90:365-394): zero fluxes, open-water friction velocity
        if layout is None:
            hostinit.init_forcing(gpu, case)
        else:
            gpu.set("area", tile_area)                     # (the fields came with the window of the whole domain)
        gpu.set("full_physics", 1)
        if not args.frozen_diffusivities:
            # difest_isobml's diffusivity estimates live (stage_difest_iso.hip): NorESM's &DIFFUSION defaults for this coordinate
            if layout is None:
                hostinit.init_difest(gpu, case, device=True)
            else:
                import math
                gpu.set("bdml_logc", math.log(2. * hostinit._BVF0 / hostinit._CORI30))
            for d_ in hostinit.DIFEST_NORESM:
                for nm, v in d_.items():
                    gpu.set(nm, v)
            gpu.set("difest_live", 1)
            difest_live = True
    for o in args.opt:
        nm, v = o.split("=")
        gpu.set(nm, int(v))
    baclin = case.params["baclin"]

    # ---- warm-up: first (forward) step + W-1 leap-frog steps, in the mode the timed steps run in: plain launches without the stage
    # timers (stage times come from the 5 steps after the timed region).  With the option use_graph = 1 -- off by default: replay measures
    # 2 % SLOWER than plain launches on ROCm 7.2 -- blomgpu_step captures the graphs of both time-level parities in its fourth step from rest, so a
    # warm-up of W >= 4 steps would leave nothing but replays to the timed region; `timed_steps_replayed_as_graphs` in the line says how many were.
    ns = gpu.step(0, args.spinup) if args.spinup > 0 else 0
    ns = gpu.step(ns, args.warmup) if args.warmup > 0 else ns
    gpu.sync()
    classes = ["cmnfld", "difest", "eddtra", "remap", "cppm", "diffus", "pgforc", "momtum", "convec", "diapfl", "thermf", "mxlayr", "barotp",
               "pbcor1", "pbcor2", "tmsmt", "init_fluxes", "updtrc"]

    # ---- timed region --------------------------------------------------------------------------
    def barrier():
        if world > 1:
            torch.distributed.barrier(device_ids=[local]) if on_gpu else torch.distributed.barrier()
    barrier()
    gpu.sync()
    cuda_sync()
    g0 = gpu.get_real("graph_steps")
    t0 = time.perf_counter()
    ns = gpu.step(ns, args.steps)
    gpu.sync()
    cuda_sync()
    barrier()
    dt = time.perf_counter() - t0
    graph_steps_timed = int(gpu.get_real("graph_steps") - g0)     # how many of the timed steps were graph replays (0 unless --opt use_graph=1)
    dt = launch.max_over_ranks(dt, env, device="cuda" if on_gpu else "cpu")

    # ---- dominant kernel class, timed with HIP events on the library's stream over K more steps
    gpu.set("timing", 1)
    gpu.timer_reset()
    ns = gpu.step(ns, min(args.steps, 5))
    gpu.sync()
    live = {}
    for cl in classes:
        ms, n = gpu.timer_get(cl)
        if n:
            live[cl] = ms / max(1, min(args.steps, 5))
    kern = {}
    for kn in ("k_remap_tile", "k_mom_cor_march", "k_mom_visc_march", "k_diapfl_column3", "k_bt_steps", "k_pgf_uv", "k_pbc_tile"):
        ms, n = gpu.timer_get(kn)
        if n:
            kern[kn] = (ms / n, n / max(1, min(args.steps, 5)))          # average launch duration [ms], launches per step
    exch = gpu.timer_get("exchange")
    gpu.set("timing", 0)
    exch_all = [round(x, 4) for x in launch.all_gather_objects(exch[0] / max(1, min(args.steps, 5)), env)] if layout is not None else None
    import numpy as np
    finite = bool(np.isfinite(gpu.get("u")).all() and np.isfinite(gpu.get("dp")).all())
    crc_state = gpu.crc("dp", 1, 2 * case.kdm, 1) ^ gpu.crc("u", 1, 2 * case.kdm, 3) if layout is None else None
    # (the checksums of the state the TIMED steps left -- taken before the comparison steps below, on tiles as on one tile, so that
    # state_crc is the same for every N at equal --steps/--warmup)
    my_strips = {nm: gpu.crc_strips(nm, 1, 2 * case.kdm, it) for nm, it in (("dp", 1), ("u", 3))} if layout is not None else None
    # ---- blocks 2..B: the same number of steps timed again, as the headline block was (round 5's review: the timed region is 0.14 s
    # and the spread between boxes larger than the changes of a round; median / min / max over the blocks say what a number resolves)
    def timed_block(nsteps):
        nonlocal ns
        barrier()
        gpu.sync()
        cuda_sync()
        t1 = time.perf_counter()
        ns = gpu.step(ns, nsteps)
        gpu.sync()
        cuda_sync()
        barrier()
        return launch.max_over_ranks(time.perf_counter() - t1, env, device="cuda" if on_gpu else "cpu") / nsteps * 1e3
    block_ms = [dt / args.steps * 1e3] + [timed_block(args.steps) for _ in range(max(0, args.blocks - 1))]
    # ---- a state a model run is in: the run continued to --spunup-steps, then --steps steps timed again and the classes once more
    spunup = None
    if full and world == 1 and layout is None and args.spunup_steps > ns:
        ns = gpu.step(ns, args.spunup_steps - ns)
        at = ns
        sp_blocks = [timed_block(args.steps) for _ in range(3)]
        gpu.set("timing", 1)
        gpu.timer_reset()
        ns = gpu.step(ns, min(args.steps, 5))
        gpu.sync()
        sp_stages = {}
        for cl in classes:
            ms, n = gpu.timer_get(cl)
            if n:
                sp_stages[cl] = ms / max(1, min(args.steps, 5))
        gpu.set("timing", 0)
        spunup = {"after_steps": at, "ms_per_step": sorted(sp_blocks)[1], "ms_per_step_blocks": [round(x, 4) for x in sp_blocks],
                  "value": baclin / 86400.0 / (sorted(sp_blocks)[1] * 1e-3), "stages_ms": sp_stages,
                  "state_finite": bool(np.isfinite(gpu.get("u")).all() and np.isfinite(gpu.get("dp")).all()),
                  "note": "the same run continued: median of three blocks of --steps steps from this step count on (the headline is timed "
                          "from rest, in the transient of the mixed layer: profiles/r06_longrun.txt)"}
    dyncore_ms = None
    if full and not args.no_dyncore_compare:
        # the dynamical core alone (the sequence rounds 1-3 timed), in the same run on the same device, for comparison
        gpu.set("full_physics", 0)
        ns = gpu.step(ns, 3)
        gpu.sync()
        cuda_sync()
        t1 = time.perf_counter()
        ns = gpu.step(ns, args.steps)
        gpu.sync()
        cuda_sync()
        dyncore_ms = (time.perf_counter() - t1) / args.steps * 1e3
    finite = all(launch.all_gather_ints(int(finite), env))
    if layout is not None:
        # xccrc of the whole domain (phy/mod_xc.F90:2195-2322) chained over the tiles: equal to the single tile's
        from blom_amd.tiles import chain_crc
        parts = launch.all_gather_objects(my_strips, env)
        tiles_of = {layout.rank_tile(r): p for r, p in enumerate(parts)}
        crcs = [chain_crc({k: v["dp"] for k, v in tiles_of.items()}, layout) ^ chain_crc({k: v["u"] for k, v in tiles_of.items()}, layout)]
    else:
        # weak scaling: every tile integrates the same periodic pattern, so all ranks must hold the same bits
        crcs = launch.all_gather_ints(crc_state, env)

    ms_per_step = dt / args.steps * 1e3
    # model days per wall second of the domain that is integrated (for --scaling weak that domain is N times as long:
    # tile_days_per_s = N x value is reported beside it, not as the value)
    value = args.steps * baclin / 86400.0 / dt
    tile_i, tile_j = (layout.tile(*layout.rank_tile(rank))[2:] if layout is not None else (case.idm, case.jdm))
    F = tile_i * tile_j * case.kdm * 8.0
    cb = class_bytes_F(case.ntr, ntr_diffused(case))
    hbm_classes = {k: v for k, v in live.items() if k in cb}
    dom = max(hbm_classes, key=hbm_classes.get)
    a3d, a2d = algorithmic_bytes(case, case.ntr)
    kb = kernel_bytes_F(case.ntr, sum(1 for nt in range(1, case.ntr + 1) if not (case.params.get("itrtke", -1) >= 1 and not case.params.get("tkeadv", 1)
                                                                               and nt in (case.params.get("itrtke"), case.params.get("itrgls")))))
    traffic, traffic_src = class_traffic(args.config, case.ntr)
    out = {
        "metric": "simulated-days/sec", "value": value, "unit": "simulated-days/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{args.config} {case.idm}x{case.jdm}x{case.kdm} as {world} tiles ({layout.npx}x{layout.npy}: columns "
                                f"{'/'.join(map(str, layout.isizes))}, rows {'/'.join(map(str, layout.jsizes))}), 1 tile per GPU, "
                                if layout is not None else
                                f"{args.config} {case.idm * world}x{case.jdm}x{case.kdm} as {world} tile(s) of "
                                f"{case.idm}x{case.jdm}x{case.kdm} along i, 1 tile per GPU, ") +
                               f"isopyc_bulkml/{args.advmth}/geopotential/uc/enscon, ntr={case.ntr} "
                               f"({'TKE, length-scale slot, ideal age: the reference default build' if case.ntr == 3 else 'ideal age' if case.ntr == 1 else f'the default three + {case.ntr - 3} passive tracers'}), "
                               f"baclin={baclin:g}s batrop={case.params['batrop']:g}s lstep={case.params['lstep']}; " +
                               (f"config 2's step (phy/mod_blom_step.F90:96-253: init_fluxes, tmsmt1, cmnfld2, difest_isobml "
                                f"[{'the whole routine: diffusivities estimated every step (NorESM defaults incl. rhsctp; the topographic beta is a synthetic field)' if difest_live else 'halos, pressure, ustar3, niw_ke_tendency; diffusivities frozen'}], eddtra, advect, pbcor1, "
                                f"diffus, pgforc, momtum, convec, diapfl, thermf, mxlayr, updtrc, barotp, pbcor2, tmsmt2, cmnfld1), the channel "
                                f"experiment's own forcing (zero fluxes, ustarw = 0.005 m/s), (gm, " if full else
                                f"full dyncore stage sequence incl. cmnfld2, eddtra and convec (gm, ") +
                               (f"neutral slopes from cmnfld2 every step" if args.slopes == "live" else f"frozen slopes of amplitude {NSLP0:g}") + "); "
                               "N>1: halos over RCCL send/recv" + (f", barotropic solve {args.barotp}" if layout is not None else "") + "; state_crc = xccrc(dp) ^ xccrc(u) of the whole domain, "
                               "the same for every N at equal --steps/--warmup" + (" [halo via RCCL self-send]" if args.rccl_self else ""),
                   "eddtra_parity": "unpinned (mod_eddtra needs CVMix: the reference build lacks it; checked against the C restatement)",
                   "state_finite": finite, "tiles_bit_identical": len(set(crcs)) == 1,
                   "state_crc": f"{crcs[0]:08x}"},
        "roofline": roofline_of_kernel(kern, kb, F, args.config) or
                    {"bound": "hbm", "kernel": dom, "achieved": cb[dom] * F / (live[dom] * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": cb[dom] * F / (live[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": (traffic or {}).get(dom), "traffic_source": traffic_src,
                     "algorithmic_bytes": cb[dom] * F, "avg_ms": live[dom]},
        # the stage class with the largest time against the bytes SURVEY.md 8(d) gives the class (rounds 1-3 reported this one)
        "class_roofline": {"bound": "hbm", "class": dom, "achieved": cb[dom] * F / (live[dom] * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": cb[dom] * F / (live[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": (traffic or {}).get(dom), "traffic_source": traffic_src,
                     "algorithmic_bytes": cb[dom] * F, "avg_ms": live[dom],
                     # the 3-D (HBM) part of the step alone: A3D / step time against the peak -- A2D, barotp's 2-D working set,
                     # stays on chip and is left out of this one
                     "step_hbm_frac": a3d / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * (world if layout is not None else 1)),
                     "dominant_single_kernel": dominant_kernel(args.config)},
        "step_roofline": {"A3D_bytes": a3d, "A2D_bytes": a2d,
                          # barotp's substep loop: the 2-D working set it sweeps per step (A2D, SURVEY.md 8d: 62 planes per substep) over
                          # the time of its k_bt_steps launches -- traffic that stays in registers, LDS and L2, not an HBM figure
                          "barotp_onchip": ({"kernel": "k_bt_steps", "launches_per_step": kern["k_bt_steps"][1],
                                             "ms_per_step": kern["k_bt_steps"][0] * kern["k_bt_steps"][1],
                                             "A2D_GBs": a2d / (kern["k_bt_steps"][0] * kern["k_bt_steps"][1] * 1e-3) / 1e9}
                                            if "k_bt_steps" in kern else None),
                          "achieved_GBs": (a3d + a2d) / (ms_per_step * 1e-3) / 1e9,
                          "frac": (a3d + a2d) / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * (world if layout is not None else 1)),
                          # HBM bytes the PMC counters saw per step in the committed profile of this workload (all stage
                          # classes), and the rate they correspond to at this run's step time
                          "counted_traffic_bytes": sum(traffic.values()) if traffic else None,
                          "counted_traffic_GBs": (sum(traffic.values()) / (ms_per_step * 1e-3) / 1e9) if traffic else None},
        "stages_ms": live,
        # counted HBM bytes (committed PMC profile of this workload: 2 x FETCH_SIZE + WRITE_SIZE) over the algorithmic bytes of SURVEY.md 8(d),
        # per stage class that has both
        "class_counted_over_alg": ({k: round(traffic[k] / (cb[k] * F), 2) for k in cb if traffic and k in traffic and k in live} or None),
    }
    out["config"]["physics"] = "full" if full else "dyncore"
    out["config"]["forcing"] = args.forcing
    out["config"]["spinup_steps"] = args.spinup
    srt = sorted(block_ms)
    out["ms_per_step_blocks"] = [round(x, 4) for x in block_ms]
    out["timed_steps_replayed_as_graphs"] = graph_steps_timed
    out["ms_per_step_median"] = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    out["ms_per_step_min"], out["ms_per_step_max"] = srt[0], srt[-1]
    if spunup is not None:
        out["spunup"] = spunup
    if dyncore_ms is not None:
        out["dyncore_only"] = {"ms_per_step": dyncore_ms, "value": baclin / 86400.0 / (dyncore_ms * 1e-3),
                               "note": "the dynamical-core sequence of rounds 1-3 (no thermf, mxlayr, difest part, cmnfld1), timed in this run after the main measurement"}
    if world > 1 and layout is None:
        out["tile_days_per_s"] = world * value
    if layout is not None:
        # the Amdahl term of the strong-scaling design: with --barotp replicated every rank solves the WHOLE 2-D barotropic domain
        # (no exchange inside the substep loop), so this part of a step does not shrink with N; rank_share_ms is what does
        bt = live.get("barotp")
        out["strong_scaling_terms"] = {"barotp": args.barotp, "barotp_replicated_ms": bt if args.barotp == "replicated" else None,
                                       "barotp_ms": bt, "rank_share_ms": (ms_per_step - bt) if bt is not None else None,
                                       "exchange_ms_per_rank": exch_all,
                                       "note": "HIP-event time of the barotp class on this rank (rank 0), step time minus it; exchange_ms_per_rank: "
                                               "per step and rank, HIP events around every pack + send/recv + unpack on the stream it goes to (halo "
                                               "updates, the arctic strips, the gather of the replicated barotropic solve), from the K steps "
                                               "timed by class after the main measurement -- a rank that waits for a slower peer shows it here"}
    if rank == 0:
        # stdout carries the ONE JSON line and nothing else: the reference library prints through the Fortran
        # runtime (bigrid messages, buffered unit 6 flushed at exit), so from here on file descriptor 1 points
        # to stderr and the line goes out through a duplicate of the original stdout
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(case.name, case, masks, nreg, live=args.slopes == "live", full=full, difest=difest_live,
                                                   start=(gpu, ns) if args.spinup > 0 else None)
            except Exception as e:                       # the bench line must still be produced
                out["cpu_baseline"] = {"error": repr(e)}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        os.close(real_stdout)
    if world > 1 or args.rccl_self or layout is not None:
        gpu.rccl_finalize()
    gpu.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
