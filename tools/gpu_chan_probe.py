import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bench import build_case
from blom_amd.gpu import BlomGpu
from blom_amd import hostinit
from blom_amd.stepper import DYNCORE_STAGES
case, nreg, masks = build_case(sys.argv[1] if len(sys.argv) > 1 else "channel")
gpu = BlomGpu(case.idm, case.jdm, case.kdm, case.ntr, nreg, masks)
print("created", flush=True)
hostinit.init_state(gpu, case); gpu.sync()
print("init done", flush=True)
m, n, mm, nn, k1m, k1n = hostinit.step_indices(0, case.kdm)
gpu.set("nstep", 1)
for st in DYNCORE_STAGES:
    gpu.stage(st, m, n, mm, nn, k1m, k1n); gpu.sync()
    print(st, "ok", flush=True)
print(hex(gpu.crc("dp", 1, 2 * case.kdm, 1)), flush=True)
print(hex(gpu.crc("u", 1, 2 * case.kdm, 3)), flush=True)
