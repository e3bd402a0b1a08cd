#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REFERENCE's own compiled code
(oracle/_ref/<cfg>/libblomref.so, built from /root/reference by oracle/Makefile).  Run in the
build container only:   python tests/golden/make_golden.py

Per configuration (chan_s, box_s; for fuk95 -- the reference's own test case, 156x32x12 -- and tri_s --
arctic patch -- only the CRC file: their inputs are the analytic host initialisation, which the tests redo):
  <cfg>_init.npz   complete model state + masks + grid after host initialisation (the inputs)
  <cfg>_crc.json   for steps 1..NSTEPS and every stage of the dyncore sequence: the reference's own
                   chksum/xccrc value (phy/mod_checksum.F90, phy/mod_xc.F90:4164) of every field
                   after that stage -- the reference's de-facto golden-vector mechanism
  <cfg>_final.npz  selected full fields at the end of the last step (expected outputs)
Fixtures are data only; no reference source is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from blom_amd.cases import make_case            # noqa: E402
from blom_amd import hostinit                    # noqa: E402
from blom_amd.checksum import grid_of            # noqa: E402
from blom_amd.stepper import dyncore_step        # noqa: E402
from oracle.refblom import get_ref_backend       # noqa: E402
from parity import STATE_FIELDS, GRID_FIELDS, INT_FIELDS   # noqa: E402

NSTEPS = 3
SCRATCH = {"uflux", "vflux", "uflux2", "vflux2", "uflux3", "vflux3", "utotm", "vtotm", "util1", "util2"}
CRC_FIELDS = [f for f in STATE_FIELDS if f not in SCRATCH]
FINAL_FIELDS = ["u", "v", "dp", "temp", "saln", "sigma", "pb", "ub", "vb", "trc"]

for cfg in ("chan_s", "box_s", "fuk95", "tri_s"):
    crc_only = cfg in ("fuk95", "tri_s")
    NSTEPS = 2 if cfg == "fuk95" else 3
    case = make_case(cfg)
    ref = get_ref_backend(cfg, case.depth)
    hostinit.init_state(ref, case)
    init = {nm: ref.get(nm).copy() for nm in STATE_FIELDS + GRID_FIELDS + INT_FIELDS if ref.ref.has_field(nm) or nm in ("trc",)}
    for m in ("ip", "iu", "iv", "iq"):
        init["mask_" + m] = ref.masks[m].copy()
    if not crc_only:
        np.savez_compressed(os.path.join(HERE, f"{cfg}_init.npz"), **init)
    crcs = {}
    state = {}

    def record(st):
        crcs.setdefault(str(state["step"]), {})[st] = {
            nm: ref.ref.xccrc(ref.get(nm), grid_of(nm)) for nm in CRC_FIELDS}

    ns = 0
    for _ in range(NSTEPS):
        state["step"] = ns + 1
        pending = []

        def hook(st, six):
            if pending:
                record(pending.pop())
            pending.append(st)
        ns = dyncore_step(ref, ns, case.params["baclin"], hook=hook)
        record(pending.pop())
    json.dump({"nsteps": NSTEPS, "fields": CRC_FIELDS, "crc": crcs},
              open(os.path.join(HERE, f"{cfg}_crc.json"), "w"))
    if not crc_only:
        np.savez_compressed(os.path.join(HERE, f"{cfg}_final.npz"), **{nm: ref.get(nm).copy() for nm in FINAL_FIELDS})
    print(cfg, "fixtures written")
