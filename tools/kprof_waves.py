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
ap.add_argument("--sel", type=int, default=1, help="which marked kernel writes the buffer: 1 k_pgf_uv*, 2 k_diapfl_column3 (blomgpu_internal.h: kprof_sel)")
ap.add_argument("--nt", type=int, default=4, help="number of timestamp words (the rest are counters)")
ap.add_argument("--config", default="channel")
ap.add_argument("--tracers", default="default")
ap.add_argument("--labels", default=None, help="comma separated names of the phases between consecutive timestamps")
args = ap.parse_args()
case, nreg, masks = bench.build_case(args.config, "remap", args.tracers)
gpu = bench.device_for_bench(case, nreg, masks, live=True)
for o in args.opt:
    nm, v = o.split("=")
    gpu.set(nm, int(v))
gpu.set("overlap", 0)
nw = 8 * args.waves
gpu.lib.blomgpu_dbg_kprof.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ns = gpu.step(0, args.spinup + args.steps - 1)
assert gpu.lib.blomgpu_dbg_kprof(gpu.ctx, None, nw) == 0
gpu.set("kprof_sel", args.sel)
gpu.set("overlap", 0)                  # (setting an option drops the captured graphs: their launches carry the old buffer pointer)
ns = gpu.step(ns, 1)
buf = np.zeros(nw, dtype=np.int64)
assert gpu.lib.blomgpu_dbg_kprof(gpu.ctx, buf.ctypes.data_as(C.c_void_p), nw) == 0
gpu.close()
w = buf.reshape(-1, 8)
w = w[w[:, 0] > 0]
if args.save:
    np.save(args.save, w)
nt = args.nt
t = w[:, :nt].astype(np.float64) * 0.01            # us
for s_ in range(1, nt):                              # a phase a column skips leaves its word at zero: it takes the previous mark's time
    t[:, s_] = np.where(w[:, s_] > 0, t[:, s_], t[:, s_ - 1])
t0 = t[:, 0].min()
span = t[:, nt - 1].max() - t0
life = t[:, nt - 1] - t[:, 0]
q = lambda a: " ".join(f"{np.percentile(a, p):8.1f}" for p in (0, 10, 50, 90, 99, 100))
labels = args.labels.split(",") if args.labels else (["start .. first level", "level loop", "after the loop"] if nt == 4 else [f"phase {s_}" for s_ in range(1, nt)])
print(f"step {ns}, options {args.opt}, kernel {args.sel}: {len(w)} waves; launch span {span:.1f} us (first start to last end); starts spread over {t[:, 0].max() - t0:.1f} us")
print("                               min      p10      p50      p90      p99      max   [us]")
print(f"{'wave lifetime':26s} {q(life)}   mean {life.mean():.1f}")
for s_ in range(1, nt):
    d_ = t[:, s_] - t[:, s_ - 1]
    print(f"{labels[s_ - 1]:26s} {q(d_)}   mean {d_.mean():.1f}")
print(f"{'end time - first start':26s} {q(t[:, nt - 1] - t0)}")
# the waves that end last: where their time went
late = np.argsort(t[:, nt - 1])[-max(1, len(w) // 50):]
print("the 2 % of the waves that end last, mean of their phases:", " ".join(f"{labels[s_ - 1]} {np.mean(t[late, s_] - t[late, s_ - 1]):.1f}" for s_ in range(1, nt)))
for s_ in range(nt, 8):
    if w[:, s_].any():
        c_ = np.corrcoef(life, w[:, s_])[0, 1] if w[:, s_].std() > 0 else float("nan")
        print(f"word {s_} (counter, summed over the lanes): mean {w[:, s_].mean():.1f}, max {w[:, s_].max()}, correlation with the wave's lifetime {c_:.2f}; mean over the late waves {w[late, s_].mean():.1f}")
