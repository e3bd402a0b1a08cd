"""Per-wavefront phase timestamps of a column kernel (GPU box; library built with `make -C blom_amd/csrc kprof` ->
tools/probes/libblomgpu_kprof.so, loaded through BLOMGPU_LIB).  Runs the bench workload for --steps steps (+ --spinup), reads the
buffer the LAST launch of the marked kernel filled (8 words a wave: t0 start, t1 after the first level, t2 after the level loop, t3
end [100 MHz ticks], word 4 / 5: event counts summed over the lanes), and prints how the launch's duration is made up: the span
from the first start to the last end, the distribution of the waves' lifetimes and of their phases.

usage: BLOMGPU_LIB=tools/probes/libblomgpu_kprof.so python3 tools/kprof_waves.py [--steps 12] [--spinup 0] [--opt name=int ...] [--waves N]"""
import argparse
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--spinup", type=int, default=0)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--waves", type=int, default=4096)
ap.add_argument("--save", default=None)
args = ap.parse_args()
case, nreg, masks = bench.build_case("channel", "remap", "default")
gpu = bench.device_for_bench(case, nreg, masks, live=True)
for o in args.opt:
    nm, v = o.split("=")
    gpu.set(nm, int(v))
gpu.set("overlap", 0)
nw = 8 * args.waves
gpu.lib.blomgpu_dbg_kprof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ns = gpu.step(0, args.spinup + args.steps - 1)
assert gpu.lib.blomgpu_dbg_kprof(gpu.ctx, None, nw) == 0
gpu.set("overlap", 0)                  # (setting an option drops the captured graphs: their launches carry the old buffer pointer)
ns = gpu.step(ns, 1)
buf = np.zeros(nw, dtype=np.int64)
assert gpu.lib.blomgpu_dbg_kprof(gpu.ctx, buf.ctypes.data_as(C.c_void_p), nw) == 0
gpu.close()
w = buf.reshape(-1, 8)
w = w[w[:, 0] > 0]
if args.save:
    np.save(args.save, w)
t = w[:, :4].astype(np.float64) * 0.01            # us
t0 = t[:, 0].min()
span = t[:, 3].max() - t0
life = t[:, 3] - t[:, 0]
q = lambda a: " ".join(f"{np.percentile(a, p):8.1f}" for p in (0, 10, 50, 90, 99, 100))
print(f"step {ns}, options {args.opt}: {len(w)} waves; launch span {span:.1f} us (first start to last end); starts spread over {t[:, 0].max() - t0:.1f} us")
print("                         min      p10      p50      p90      p99      max   [us]")
print(f"wave lifetime        {q(life)}   mean {life.mean():.1f}")
print(f"start .. first level {q(t[:, 1] - t[:, 0])}   mean {(t[:, 1] - t[:, 0]).mean():.1f}")
print(f"level loop           {q(t[:, 2] - t[:, 1])}   mean {(t[:, 2] - t[:, 1]).mean():.1f}")
print(f"after the loop       {q(t[:, 3] - t[:, 2])}   mean {(t[:, 3] - t[:, 2]).mean():.1f}")
print(f"end time - first start {q(t[:, 3] - t0)}")
print(f"word 4 (moves, summed over lanes) mean {w[:, 4].mean():.1f}   word 5 (levels x lanes moving by more than one layer) mean {w[:, 5].mean():.1f}")
c = np.corrcoef(life, w[:, 5])[0, 1] if w[:, 5].std() > 0 else float('nan')
c4 = np.corrcoef(life, w[:, 4])[0, 1] if w[:, 4].std() > 0 else float('nan')
print(f"correlation of a wave's lifetime with word 5: {c:.2f}, with word 4: {c4:.2f}")
