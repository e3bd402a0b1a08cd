#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of the built library: reads the AMDGPU metadata notes of the
device code objects inside blom_amd/lib/libblomgpu.so (or the .o files given) and prints one line per kernel
with the wavefronts per SIMD its VGPR count allows on gfx950 (512 VGPRs per lane and SIMD, granule 8)."""
import os, re, subprocess, sys, tempfile, glob
LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def notes_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fb.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb], capture_output=True)
        out = os.path.join(td, "dev.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(out) or os.path.getsize(out) == 0:
            return ""
        return subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout

def parse(txt):
    ks = []
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda key: (re.search(r"\." + key + r":\s*(\S+)", blk) or [None, "0"])[1]
        name = g("name")
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")
        ks.append((name, int(g("vgpr_count")), int(blk.split()[0]), int(g("sgpr_count")), int(g("private_segment_fixed_size")),
                   int(g("group_segment_fixed_size")), int(g("max_flat_workgroup_size")), int(g("vgpr_spill_count")), int(g("sgpr_spill_count"))))
    return ks

if __name__ == "__main__":
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "blom_amd/csrc/build/*.o")))
    print(f"{'kernel':58s} vgpr agpr sgpr scratch   lds  wgmax vspill sspill waves/SIMD")
    for o in objs:
        for k in parse(notes_of(o)):
            tot = k[1] + k[2]
            gran = (tot + 7) // 8 * 8
            waves = min(8, 512 // max(gran, 8))
            print(f"{k[0][:58]:58s} {k[1]:4d} {k[2]:4d} {k[3]:4d} {k[4]:7d} {k[5]:6d} {k[6]:5d} {k[7]:6d} {k[8]:6d} {waves:5d}")
