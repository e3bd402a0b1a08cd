#!/usr/bin/env python3
"""Benchmark of the MI355X-native BLOM dynamical core.

  python bench.py --gpus N --steps K --warmup W

A "step" is one baroclinic time step of the hot path (stage sequence of
phy/mod_blom_step.F90:96-253 restricted to the dynamical core: init_fluxes, tmsmt1, [halo updates
of cmnfld2/difest], advect(remap), pbcor1, diffus, pgforc, momtum, diapfl, [mxlayr dp-halo tail],
barotp, pbcor2, tmsmt2) on the `channel` configuration of BASELINE.json (208x512x53, 1 tile per
GPU; for N>1 the channel is N times as long and the tiles exchange halos over RCCL), with the
state resident in HBM.  metric = simulated model days per wall second
= steps/s * baclin / 86400.

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


def build_case(cfg, advmth="remap", tracers="default"):
    import numpy as np
    from blom_amd.cases import make_case
    from blom_amd import hostinit
    # "default": the reference's default build options (meson_options.txt:17-21: TKE + advection of it, ideal age)
    # => ntr = 3, the tracer count SURVEY.md 8(d) quotes the channel on; "iage": -DTRC -DIDLAGE only, ntr = 1
    case = make_case(cfg + ("_tke" if tracers == "default" else ""), nslp0=NSLP0, advmth=advmth)
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


def class_traffic(config, ntr=1):
    """HBM bytes per step and kernel class from the newest committed PMC profile of this configuration
    (profiles/*_class_traffic.json, written by tools/prof_summarize.py from separate rocprofv3 --pmc
    passes of this very command); None when there is none."""
    import glob
    import json
    if config != "channel":
        return None, None
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_class_traffic.json")))
    if not files:
        return None, None
    for f in reversed(files):
        d = json.load(open(f))
        if d.get("ntr", 1) == ntr:                      # a profile of this very workload
            return d.get("bytes_per_step", {}), os.path.basename(f)
    return None, None


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


def cpu_baseline(cfg, case, masks, nreg, max_seconds=20.0):
    """Runs _cpu_baseline in a thread with a 2 GiB stack: the reference keeps its stage-local
    2-D work arrays (21 in remap, ~30 in momtum) on the stack, which at channel size exceeds the
    default 8 MiB limit (BLOM is normally run with `ulimit -s unlimited`)."""
    import threading
    res = {}
    threading.stack_size(2 << 30)
    th = threading.Thread(target=lambda: res.update(_cpu_baseline(cfg, case, masks, nreg, max_seconds)))
    th.start()
    th.join()
    threading.stack_size(0)
    return res


def _cpu_baseline(cfg, case, masks, nreg, max_seconds=20.0):
    """Reference (preferred) or C restatement timed on the host for a bounded number of steps."""
    from blom_amd import hostinit
    from blom_amd.stepper import dyncore_step
    kind = None
    cores = 1
    try:
        from oracle.refblom import get_ref_backend, have_ref
        if have_ref(cfg + "_omp"):          # the reference with its OpenMP directives on, all host cores
            cores = usable_cores()
            os.environ["OMP_NUM_THREADS"] = str(cores)
            os.environ.setdefault("OMP_PROC_BIND", "close")
            os.environ["OMP_STACKSIZE"] = "1G"   # the stages keep private 2-D work arrays on the thread stacks
            be = get_ref_backend(cfg + "_omp", case.depth)
            kind = "reference"
        elif have_ref(cfg):
            be = get_ref_backend(cfg, case.depth)
            kind = "reference"
    except Exception:
        kind = None
        cores = 1
    if kind is None:
        from oracle.coracle import COracle
        be = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
        kind = "port"
    hostinit.init_state(be, case)
    ns = dyncore_step(be, 0, case.params["baclin"])          # forward first step (untimed)
    # per-stage host times beside the device's stages_ms: the hook fires before every stage
    per_stage, mark = {}, [None, 0.0]

    def hook(st, six):
        now = time.perf_counter()
        if mark[0] is not None:
            per_stage[mark[0]] = per_stage.get(mark[0], 0.0) + (now - mark[1])
        mark[0], mark[1] = st, now
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < max_seconds and n < 200):
        ns = dyncore_step(be, ns, case.params["baclin"], hook=hook)
        hook(None, None)
        n += 1
    dt = (time.time() - t0) / n
    note = ""
    if kind == "reference":
        # mod_eddtra is not part of the reference build (it needs mod_difest -> CVMix): time that one
        # stage on the C restatement, on the same inputs, and add it
        from oracle.coracle import COracle
        co = COracle(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
        hostinit.init_state(co, case)
        six = hostinit.step_indices(1, case.kdm)
        co.set("delt1", 2 * case.params["baclin"])
        co.stage("eddtra", *six)
        t1 = time.time()
        for _ in range(3):
            co.stage("eddtra", *six)
        de = (time.time() - t1) / 3
        dt += de
        per_stage["eddtra"] = de * n
        note = f"; eddtra ({de * 1e3:.1f} ms) timed on the C restatement since the reference build lacks it"
    how = (f"{cores} OpenMP threads (reference built with -fopenmp)" if cores > 1 else
           "single thread (reference built without OpenMP)" if kind == "reference" else "single thread (C restatement)")
    stages_ms = {k: round(v / n * 1e3, 2) for k, v in per_stage.items() if k}
    # the device reports advect as "remap" (or "cppm"): same name here
    if "advect" in stages_ms:
        stages_ms[case.params.get("advmth", "remap")] = stages_ms.pop("advect")
    return dict(value=case.params["baclin"] / 86400.0 / dt, unit="simulated-days/sec", cores=cores, kind=kind, stages_ms=stages_ms,
                sample=f"{n} baroclinic steps of the same {cfg} workload, {dt * 1e3:.1f} ms/step, {how}{note}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="channel")
    ap.add_argument("--advmth", default="remap", choices=["remap", "cppm"],
                    help="advection method (the reference's advmth); the headline configuration is remap")
    ap.add_argument("--tracers", default="default", choices=["default", "iage"],
                    help="default: the reference's default option set (TKE, its advection, ideal age: ntr = 3); "
                         "iage: ideal age only (ntr = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                    help="library option for A/B runs of kernel variants, e.g. momtum_v=1 (default: production kernels)")
    ap.add_argument("--rccl-self", action="store_true",
                    help="N=1 only: route the halo update through the RCCL transport (rank sends to itself) "
                         "to measure the exchange overhead of the N>1 path on one GPU")
    args = ap.parse_args()

    from blom_amd import launch
    env = launch.rank_env()
    rank, world, local = env.rank, env.world, env.local
    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from blom_amd.gpu import BlomGpu, rccl_unique_id
    from blom_amd import hostinit
    case, nreg, masks = build_case(args.config, args.advmth, args.tracers)
    if world > 1:
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
    hostinit.init_state(gpu, case)
    for o in args.opt:
        nm, v = o.split("=")
        gpu.set(nm, int(v))
    baclin = case.params["baclin"]

    # ---- warm-up: first (forward) step + W-1 leap-frog steps, with per-class HIP-event timing ----
    ns = gpu.step(0, 1)
    gpu.set("timing", 1)
    gpu.timer_reset()
    if args.warmup > 1:
        ns = gpu.step(ns, args.warmup - 1)
    gpu.sync()
    classes = ["eddtra", "remap", "cppm", "diffus", "pgforc", "momtum", "convec", "diapfl", "barotp", "pbcor1", "pbcor2"]
    stage_ms = {}
    for cl in classes:
        ms, n = gpu.timer_get(cl)
        if n:
            stage_ms[cl] = ms / n
    gpu.set("timing", 0)

    # ---- timed region --------------------------------------------------------------------------
    def barrier():
        if world > 1:
            torch.distributed.barrier(device_ids=[local])
    barrier()
    gpu.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ns = gpu.step(ns, args.steps)
    gpu.sync()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    dt = launch.max_over_ranks(dt, env, device="cuda")

    # ---- dominant kernel class, timed with HIP events on the library's stream over K more steps
    gpu.set("timing", 1)
    gpu.timer_reset()
    ns = gpu.step(ns, min(args.steps, 5))
    gpu.sync()
    live = {}
    for cl in classes:
        ms, n = gpu.timer_get(cl)
        if n:
            live[cl] = ms / n
    gpu.set("timing", 0)
    import numpy as np
    finite = bool(np.isfinite(gpu.get("u")).all() and np.isfinite(gpu.get("dp")).all())
    # every tile integrates the same periodic pattern, so all ranks must hold the same bits
    crcs = launch.all_gather_ints(gpu.crc("dp", 1, 2 * case.kdm, 1) ^ gpu.crc("u", 1, 2 * case.kdm, 3), env)

    ms_per_step = dt / args.steps * 1e3
    # units all ranks processed / time: model days of one 208x512x53-sized tile, times the number
    # of tiles (the N-GPU job integrates an N-times longer channel at the same days/s)
    value = world * args.steps * baclin / 86400.0 / dt
    F = case.idm * case.jdm * case.kdm * 8.0
    cb = class_bytes_F(case.ntr, ntr_diffused(case))
    hbm_classes = {k: v for k, v in live.items() if k in cb}
    dom = max(hbm_classes, key=hbm_classes.get)
    a3d, a2d = algorithmic_bytes(case, case.ntr)
    traffic, traffic_src = class_traffic(args.config, case.ntr)
    out = {
        "metric": "simulated-days/sec", "value": value, "unit": "simulated-days/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.config} {case.idm * world}x{case.jdm}x{case.kdm} as {world} tile(s) of "
                               f"{case.idm}x{case.jdm}x{case.kdm} along i, 1 tile per GPU, "
                               f"isopyc_bulkml/{args.advmth}/geopotential/uc/enscon, ntr={case.ntr} "
                               f"({'TKE, length-scale slot, ideal age: the reference default build' if case.ntr == 3 else 'ideal age'}), "
                               f"baclin={baclin:g}s batrop={case.params['batrop']:g}s lstep={case.params['lstep']}; "
                               f"full dyncore stage sequence incl. eddtra and convec (gm, frozen slopes of amplitude {NSLP0:g}); "
                               "N>1: halos over RCCL send/recv, "
                               "value counts tile-days/s" + (" [halo via RCCL self-send]" if args.rccl_self else ""),
                   "state_finite": finite, "tiles_bit_identical": len(set(crcs)) == 1,
                   "state_crc": f"{crcs[0]:08x}"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": cb[dom] * F / (live[dom] * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": cb[dom] * F / (live[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": (traffic or {}).get(dom), "traffic_source": traffic_src,
                     "algorithmic_bytes": cb[dom] * F, "avg_ms": live[dom]},
        "step_roofline": {"A3D_bytes": a3d, "A2D_bytes": a2d,
                          "achieved_GBs": (a3d + a2d) / (ms_per_step * 1e-3) / 1e9,
                          "frac": (a3d + a2d) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "stages_ms": live,
    }
    if rank == 0:
        # stdout carries the ONE JSON line and nothing else: the reference library prints through the Fortran
        # runtime (bigrid messages, buffered unit 6 flushed at exit), so from here on file descriptor 1 points
        # to stderr and the line goes out through a duplicate of the original stdout
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(case.name, case, masks, nreg)
            except Exception as e:                       # the bench line must still be produced
                out["cpu_baseline"] = {"error": repr(e)}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        os.close(real_stdout)
    if world > 1 or args.rccl_self:
        gpu.rccl_finalize()
    gpu.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
