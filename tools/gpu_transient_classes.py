"""Which stage classes make the first steps from rest slower than the later ones: the bench workload stepped in chunks of 10 with the stage
timers on (plain launches, HIP events per class), one line of ms per step and class per chunk.  usage (GPU box): python3 tools/gpu_transient_classes.py [nchunks]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
case, nreg, masks = bench.build_case("channel", "remap", "default")
gpu = bench.device_for_bench(case, nreg, masks, live=True)
classes = ["cmnfld", "difest", "eddtra", "remap", "diffus", "pgforc", "momtum", "convec", "diapfl", "thermf", "mxlayr", "barotp", "pbcor1", "pbcor2", "tmsmt"]
ns = gpu.step(0, 5)
gpu.set("timing", 1)
print("steps      " + " ".join(f"{c:>7s}" for c in classes) + "     sum")
for ch in range(n):
    gpu.timer_reset()
    ns = gpu.step(ns, 10)
    gpu.sync()
    v = [gpu.timer_get(c)[0] / 10 for c in classes]
    print(f"{ns - 10:4d}-{ns:4d}  " + " ".join(f"{x:7.3f}" for x in v) + f" {sum(v):7.3f}")
gpu.close()
