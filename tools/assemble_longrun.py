"""profiles/r06_longrun.txt from the three runs of tools/longrun_full_physics.py that tools/gpu_round6_final.sh makes (gpurun_out/<tag>/longrun_*.txt).
usage: python3 tools/assemble_longrun.py gpurun_out/r06b > profiles/r06_longrun.txt"""
import sys
d = sys.argv[1]
print('''# Round 6: the long run of the bench workload against the reference, settled (successor of profiles/r05_longrun.txt).
#
# WHAT WAS ASKED (round-5 review): the device showed Tmin = -111 degC (live diffusivities, step 600) and Tmax = 60.4 degC (frozen, step 600)
# in single samples and a -2.8 % heat drift with zero surface flux; nobody knew whether the reference does the same or whether a device
# race hides behind bit-identical short runs.
#
# WHAT WAS DONE
#  * tools/longrun_reference.py ran the REFERENCE'S OWN MODULES (oracle/_ref/channel_tke_omp_xdf; the stage list of
#    stepper.FULL_STAGES_LIVE) for 600 steps from the bench's initial state in the build container (20 min on 8 threads), twice: with the
#    options of round 5 (rhsctp off: tests/golden/channel_tke_live_long_rhsctp0_crc.json -- the run the question was about) and with NorESM's
#    actual defaults as bench.py runs them since rhsctp is built (tests/golden/channel_tke_live_long_crc.json).  Each file holds xccrc of dp,
#    temp, saln, u, v, the tracers, difint, difdia at steps 100 .. 600; for EVERY step the extremes of T with their cells and the dp there,
#    and the dp-weighted sums of mass, heat and salt; for step 300 the heat sums of both time levels before and after every stage.
#  * tests/test_gpu_golden.py::test_600_steps_of_the_bench_workload_equal_the_reference_long_run (in the GPU suite, both option sets) and
#    tools/longrun_full_physics.py --golden (below) compare blomgpu_step's device-resident run with them.
#
# RESULT: THE REFERENCE DOES THE SAME, BIT FOR BIT.
#  * All checksum sets agree; every sampled step's extremes, their cells and the three sums agree (tables below).
#  * The -110.956 degC of step 600 (rhsctp off) is the reference's value too: cell (i, j, k) = (88, 293, 46), a layer with dp = 0.0 -- a MASSLESS
#    layer.  Every extreme sample of either run sits in a layer with dp between 0 and 1e-10 Pa (the reference's own traces have Tmax = 139 degC at
#    step 6, 235 degC at step 367, Tmin = -65 degC at step 349 with rhsctp off, -95 / +406 degC with it on, all at dp <= 1e-10: remap divides the
#    flux divergence of a layer by dp + 1e-12, phy/mod_remap.F90:1471-1480, and a massless layer's temperature carries no heat).  Whenever the
#    extreme cell of a step has mass (dp > 1 Pa) its value lies inside the initial range [-4.2203, 22.9].  Round 5's "60.4 degC" of the frozen
#    run is the same thing.
#  * The heat drift is the time filter's, in the reference as on the device: of the -2.3e-5 per step, tmsmt2 (phy/mod_tmsmt.F90:281-350)
#    changes the dp-weighted heat of time level m by -9.5e-5 in step 300 and pbcor2 by +4.2e-7, pbcor1 that of level n by +1.2e-7; every
#    other stage conserves it to rounding (table below; the same sums in the reference's run are equal to the last bit).  The filter
#    averages T dp over three time levels whose heat differs by ~1e-3 between odd and even steps in this state (a leapfrog
#    computational mode fed by the 0.5 m/s friction velocity) -- the reference's arithmetic, not a property of these kernels.
#  * The two maxitr counters are named as the reference's messages now (round 5 had them swapped): `mxlayr_maxitr_detrain` counts
#    'reached maxitr when detraining' (phy/mod_mxlayr.F90:440), `..._entrain` 'reached maxitr when entraining' (:950), each as the
#    reference prints it (nitr == maxitr).  All 106 080 wet columns report it in the first step (the initial mixed layer of 20 m under
#    ustar = 0.5 m/s), none afterwards: 530.4 per step averaged over the first 200 steps is that one step.
#  * A second forcing set is published: `--forcing calm`, ustarw = 5e-5 (0.005 m/s after thermf_channel's factor 1e2,
#    channel/mod_thermf_channel.F90:259): the mixed layer stays at 13 - 22 m on average (table at the end).  The headline keeps the
#    reference's own value (channel/mod_channel.F90:365, ustarw = 0.005).
#''')
print("# ---- NorESM's defaults (rhsctp on), 1 200 steps sampled every 200 (python3 tools/longrun_full_physics.py --steps 1200 --every 200 --golden tests/golden/channel_tke_live_long_crc.json) ----")
sys.stdout.write(open(f"{d}/longrun_default.txt").read())
print("#\n# ---- rhsctp off (round 5's options): steps 588 .. 600 one by one and the heat sums around every stage of step 300 (... --rhsctp 0 --steps 600 --every 1 --from 588 --budget-step 300 --golden tests/golden/channel_tke_live_long_rhsctp0_crc.json) ----")
sys.stdout.write(open(f"{d}/longrun_rhsctp0.txt").read())
print("#\n# ---- forcing calm (ustarw = 5e-5), 1 200 steps ----")
sys.stdout.write(open(f"{d}/longrun_calm.txt").read())
