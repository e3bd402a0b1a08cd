#!/usr/bin/env python3
"""Where a kernel waits for memory: disassembles the gfx950 code object inside an object file of blom_amd/csrc/build and prints, for
the kernels whose (demangled) name contains the given substring, the sequence of vector-memory instructions, s_waitcnt's and
branches in program order, run-length compressed -- e.g. `12 x global_load  |  s_waitcnt vmcnt(5)  |  <loop back to L3>`.
A `vmcnt(0)` inside a loop that also issues loads means the wave waits for its YOUNGEST load every trip (a full memory round trip
per trip, whatever was requested ahead); `vmcnt(n > 0)` waits for older loads only.

usage: tools/isa_waits.py blom_amd/csrc/build/stage_pgforc.o k_pgf_uv [--full]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def kernels(txt):
    cur, out = None, {}
    for ln in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", ln)
        if m:
            cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "")
            out[cur] = []
        elif cur is not None and ln.strip():
            out[cur].append(ln)
    return out


def summarize(lines, full=False):
    # instruction address -> index; branch targets become labels
    ins = []
    for ln in lines:
        m = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m:
            ins.append((int(m.group(2), 16), m.group(1).strip()))
    addr_idx = {a: i for i, (a, _) in enumerate(ins)}
    targets = {}
    for i, (a, t) in enumerate(ins):
        m = re.match(r"^(s_cbranch\w+|s_branch)\s+(\d+)", t)
        if m:
            off = int(m.group(2))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt in addr_idx:
                targets.setdefault(addr_idx[tgt], f"L{len(targets)}")
    ev = []

    def push(kind):
        if ev and ev[-1][0] == kind:
            ev[-1][1] += 1
        else:
            ev.append([kind, 1])
    for i, (a, t) in enumerate(ins):
        if i in targets:
            ev.append([f"{targets[i]}:", 1])
        op = t.split()[0]
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            push("load")
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
            push("store")
        elif op.startswith("global_atomic"):
            push("atomic")
        elif op.startswith("ds_"):
            if full:
                push("lds")
        elif op == "s_waitcnt":
            if "vmcnt" in t or full:
                ev.append([t, 1])
        elif op in ("s_branch",) or op.startswith("s_cbranch"):
            m = re.match(r"^(\S+)\s+(\d+)", t)
            off = int(m.group(2))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            lab = targets.get(addr_idx.get(tgt, -1), "?")
            ev.append([f"{op} -> {lab}{' (back)' if off < 0 else ''}", 1])
        elif op == "s_endpgm":
            ev.append(["s_endpgm", 1])
    return ev, len(ins)


if __name__ == "__main__":
    obj, pat = sys.argv[1], sys.argv[2]
    full = "--full" in sys.argv
    for name, lines in kernels(disassemble(obj)).items():
        if pat not in name:
            continue
        ev, n = summarize(lines, full)
        print(f"== {name}: {n} instructions")
        depth = 0
        for kind, cnt in ev:
            print("   " + (f"{cnt} x {kind}" if cnt > 1 or kind in ("load", "store", "atomic", "lds") else kind))
