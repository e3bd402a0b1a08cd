"""N > 1 host path on CPU: two real ranks over gloo (127.0.0.1) exercise the launcher helpers of
blom_amd/launch.py -- unique-id distribution, max-over-ranks timing, tile layout -- and the
message-order rule of the E/W halo exchange (blom_amd/csrc/comm_rccl.hip) against the global
xctilr of phy/mod_xc.F90:4374-4419 for periodic (west == east == the other rank) and closed
domains."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blom_amd import launch
from blom_amd.hostinit import xctilr_np

IDM, JDM, NLEV = 12, 10, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env = launch.rank_env()
        assert (env.rank, env.world, env.local) == (rank, world, rank)
        res = {}
        uid = launch.share_unique_id(lambda: bytes((7 * i + 3) % 256 for i in range(128)) if rank == 0 else b"", env)
        res["uid"] = uid
        res["tmax"] = launch.max_over_ranks(1.5 + rank, env)
        res["crcs"] = launch.all_gather_ints(1000 + rank, env)
        lay = launch.tile_layout(IDM, rank, world)
        res["layout"] = lay
        for nreg, mh, nh in ((1, 3, 3), (1, 1, 2), (0, 2, 2), (3, 4, 1), (4, 2, 3)):
            periodic = nreg not in (0, 4)
            rng = np.random.default_rng(5)                      # same global array on every rank
            G = rng.standard_normal((NLEV, JDM + 8, IDM * world + 8))
            want = G.copy()
            xctilr_np(want, 1, NLEV, mh, nh, nreg, IDM * world, JDM)
            i0 = lay["i0"]
            a = G[:, :, i0:i0 + IDM + 8].copy()
            a[:, :, :4] = np.nan                                # halos unknown before the update
            a[:, :, 4 + IDM:] = np.nan
            xctilr_np(a, 1, NLEV, 0, nh, nreg, IDM, JDM)        # phase 1: N/S is tile-local
            launch.exchange_ew_host(a, IDM, JDM, mh, nh, env, periodic)
            w = want[:, :, i0:i0 + IDM + 8]
            sel = (slice(None), slice(4 - nh, 4 + JDM + nh), slice(4 - mh, 4 + IDM + mh))
            res[f"halo{nreg}_{mh}_{nh}"] = bool(np.array_equal(a[sel], w[sel]))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    uid0 = got[0]["uid"]
    assert len(uid0) == 128
    for r in range(world):
        g = got[r]
        assert g["uid"] == uid0
        assert g["tmax"] == 1.5 + world - 1
        assert g["crcs"] == [1000 + i for i in range(world)]
        assert g["layout"] == dict(itdm=IDM * world, i0=r * IDM, px=r, npx=world)
        for k, v in g.items():
            if k.startswith("halo"):
                assert v, (r, k)


def _worker_2d(rank, npx, npy, port, q):
    world = npx * npy
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = launch.tile_layout_2d(IDM, JDM, rank, npx, npy)
        res = {"layout": lay}
        for nreg, mh, nh in ((1, 3, 3), (0, 2, 2), (3, 4, 1), (3, 2, 3), (4, 2, 3), (1, 1, 0), (4, 0, 2)):
            rng = np.random.default_rng(9)                      # same global array on every rank
            G = rng.standard_normal((NLEV, JDM * npy + 8, IDM * npx + 8))
            want = G.copy()
            xctilr_np(want, 1, NLEV, mh, nh, nreg, IDM * npx, JDM * npy)
            i0, j0 = lay["i0"], lay["j0"]
            a = G[:, j0:j0 + JDM + 8, i0:i0 + IDM + 8].copy()
            keep = a[:, 4:4 + JDM, 4:4 + IDM].copy()
            a[...] = np.nan                                     # halos unknown before the update
            a[:, 4:4 + JDM, 4:4 + IDM] = keep
            launch.exchange_2d_host(a, IDM, JDM, mh, nh, rank, npx, npy, nreg)
            w = want[:, j0:j0 + JDM + 8, i0:i0 + IDM + 8]
            # what xctilr defines: columns 1..ii of the halo rows, and all rows 1-nh..jj+nh of the halo columns
            ok = np.array_equal(a[:, 4 - nh:4 + JDM + nh, 4:4 + IDM], w[:, 4 - nh:4 + JDM + nh, 4:4 + IDM])
            for cs in (slice(4 - mh, 4), slice(4 + IDM, 4 + IDM + mh)):
                ok = ok and np.array_equal(a[:, 4 - nh:4 + JDM + nh, cs], w[:, 4 - nh:4 + JDM + nh, cs])
            res[f"halo{nreg}_{mh}_{nh}"] = bool(ok)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _worker_arctic(rank, npx, npy, port, q):
    world = npx * npy
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = launch.tile_layout_2d(IDM, JDM, rank, npx, npy)
        res = {}
        for itype in (1, 2, 3, 4, 11, 12, 13, 14):
            for mh, nh in ((3, 3), (4, 4), (2, 0), (0, 2), (1, 4)):
                rng = np.random.default_rng(itype)                  # same global array on every rank
                G = rng.standard_normal((NLEV, JDM * npy + 8, IDM * npx + 8))
                want = G.copy()
                xctilr_np(want, 1, NLEV, mh, nh, 2, IDM * npx, JDM * npy, itype=itype)   # single tile rule
                i0, j0 = lay["i0"], lay["j0"]
                a = G[:, j0:j0 + JDM + 8, i0:i0 + IDM + 8].copy()
                keep = a[:, 4:4 + JDM, 4:4 + IDM].copy()
                a[...] = np.nan
                a[:, 4:4 + JDM, 4:4 + IDM] = keep
                launch.exchange_arctic_host(a, IDM, JDM, mh, nh, rank, npx, npy, itype)
                w = want[:, j0:j0 + JDM + 8, i0:i0 + IDM + 8]
                # interior (the seam row changes in the last tile row), halo rows over 1..ii, halo columns over all rows
                ok = np.array_equal(a[:, 4 - nh:4 + JDM + nh, 4:4 + IDM], w[:, 4 - nh:4 + JDM + nh, 4:4 + IDM])
                for cs in (slice(4 - mh, 4), slice(4 + IDM, 4 + IDM + mh)):
                    ok = ok and np.array_equal(a[:, 4 - nh:4 + JDM + nh, cs], w[:, 4 - nh:4 + JDM + nh, cs])
                res[f"arctic{itype}_{mh}_{nh}"] = bool(ok)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("npx,npy", [(2, 1), (2, 2), (4, 1), (3, 1)])
def test_arctic_patch_over_gloo(npx, npy):
    """the arctic update of the RCCL transport (ordinary exchange + strips of the last tile row to all of its ranks
    + fill of the fold targets) with real ranks, for all eight grid/field types: every tile's halo -- and the seam
    row -- must be what the single-tile rule (phy/mod_xc.F90:4262-4372) gives on the global array"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_arctic, args=(r, npx, npy, port, q)) for r in range(npx * npy)]
    [p.start() for p in procs]
    out = dict(q.get(timeout=180) for _ in procs)
    [p.join(timeout=60) for p in procs]
    for r, res in out.items():
        bad = [k for k, v in res.items() if not v]
        assert not bad, (r, bad)
        assert len(res) == 40


@pytest.mark.parametrize("npx,npy", [(2, 2), (1, 2), (3, 2)])
def test_2d_tile_grid_over_gloo(npx, npy):
    """both phases of the 2-D exchange (blom_amd/csrc/comm_rccl.hip: rccl_xctilr_multi) with real ranks:
    N/S partners incl. the periodic wrap onto the same rank pair, corners travelling with the E/W strips"""
    world = npx * npy
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_2d, args=(r, npx, npy, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=180) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r in range(world):
        assert got[r]["layout"]["i0"] == (r % npx) * IDM and got[r]["layout"]["j0"] == (r // npx) * JDM
        for k, v in got[r].items():
            if k.startswith("halo"):
                assert v, (r, k)


def test_single_rank_helpers_need_no_process_group():
    env = launch.RankEnv(0, 1, 0)
    assert launch.share_unique_id(lambda: b"x" * 128, env) == b"x" * 128
    assert launch.max_over_ranks(2.5, env) == 2.5
    assert launch.all_gather_ints(7, env) == [7]
    assert launch.neighbours(0, 1, True) == (0, 0) and launch.neighbours(0, 1, False) == (-1, -1)
    assert launch.neighbours(0, 2, True) == (1, 1) and launch.neighbours(1, 2, False) == (0, -1)
    with pytest.raises(ValueError):
        launch.tile_layout(8, 2, 2)
    assert launch.neighbours_2d(0, 1, 1, True, True) == (0, 0, 0, 0)
    assert launch.neighbours_2d(3, 2, 2, True, False) == (2, 2, 1, -1)
    assert launch.neighbours_2d(4, 3, 2, False, True) == (3, 5, 1, 1)


# ---- bench.py's N > 1 path end to end, on the CPU ----------------------------------------------------------------------------
# The driver starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` on an 8-GPU node no
# session here has ever had.  This is the same command with two ranks, `--backend gloo` (no torch.cuda call) and BLOMGPU_LIB pointing
# at the host emulation of the library (tests/hostemu; its RCCL passes messages between PROCESSES through BLOM_HOSTEMU_RCCL_DIR):
# the launcher, the rendezvous, the unique-id share, the tile layout and scatter, the 2-D RCCL init, the replicated barotropic
# solve's attach, thermf's global sums on tiles, the max-over-ranks timing and the chained checksum all execute, and the state the
# two tiles end in must be the single tile's.
def _bench(tmp_path, ngpus, extra=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu = os.path.join(root, "tests", "hostemu", "libblomgpu_hostemu.so")
    if not os.path.exists(emu):
        pytest.skip("tests/hostemu/libblomgpu_hostemu.so not built")
    box = tmp_path / f"mail{ngpus}"
    box.mkdir()
    env = dict(os.environ, BLOMGPU_LIB=emu, BLOM_HOSTEMU_RCCL_DIR=str(box), OMP_NUM_THREADS="1")
    args = ["bench.py", "--gpus", str(ngpus), "--config", "chan_s", "--steps", "3", "--warmup", "2", "--backend", "gloo", "--no-cpu-baseline",
            "--no-dyncore-compare", "--blocks", "1", "--spunup-steps", "0", *extra]
    if ngpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_end_to_end_on_the_host_emulation(tmp_path):
    one = _bench(tmp_path, 1)
    two = _bench(tmp_path, 2)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["steps"] == 3
    assert two["config"]["tiles_bit_identical"] is True and two["config"]["state_finite"] is True
    assert two["config"]["physics"] == "full"                                   # config 2's step on tiles (thermf's sums on the global context)
    assert "strong_scaling_terms" in two and two["strong_scaling_terms"]["barotp"] == "replicated"
    ex = two["strong_scaling_terms"]["exchange_ms_per_rank"]             # per rank: HIP events around every pack + send/recv + unpack
    assert len(ex) == 2 and all(x > 0.0 for x in ex), ex
    # the decomposition-independent checksum: the two tiles end in the single tile's state (bench.py: state_crc)
    assert two["config"]["state_crc"] == one["config"]["state_crc"], (one["config"]["state_crc"], two["config"]["state_crc"])
