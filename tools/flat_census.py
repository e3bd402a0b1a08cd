#!/usr/bin/env python3
"""flat_* against global_* memory instructions per object file of blom_amd/csrc/build (see blomgpu_internal.h, PtrTable: a flat
access counts on vmcnt and lgkmcnt and forces `s_waitcnt vmcnt(0) lgkmcnt(0)`).  usage: tools/flat_census.py [-v]"""
import glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_waits import disassemble, kernels
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tot = [0, 0, 0, 0]
for o in sorted(glob.glob(os.path.join(ROOT, "blom_amd/csrc/build/*.o"))):
    try:
        ks = kernels(disassemble(o))
    except Exception:
        continue
    row = [0, 0, 0, 0]
    for name, lines in ks.items():
        f = sum(1 for l in lines if re.search(r"\bflat_(load|store|atomic)", l))
        g = sum(1 for l in lines if re.search(r"\bglobal_(load|store|atomic)", l))
        w0 = sum(1 for l in lines if re.search(r"s_waitcnt vmcnt\(0\)", l))
        wn = sum(1 for l in lines if re.search(r"s_waitcnt vmcnt\([1-9]\d*\)", l))
        row = [row[0] + f, row[1] + g, row[2] + w0, row[3] + wn]
        if "-v" in sys.argv and f:
            print(f"    {name[:70]:70s} flat {f:5d} global {g:5d}")
    tot = [a + b for a, b in zip(tot, row)]
    print(f"{os.path.basename(o):28s} flat {row[0]:6d}  global {row[1]:6d}  vmcnt(0) {row[2]:5d}  vmcnt(n>0) {row[3]:5d}")
print(f"{'total':28s} flat {tot[0]:6d}  global {tot[1]:6d}  vmcnt(0) {tot[2]:5d}  vmcnt(n>0) {tot[3]:5d}")
