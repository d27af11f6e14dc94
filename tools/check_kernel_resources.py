#!/usr/bin/env python3
"""Register / scratch report of every gfx950 kernel in the built objects (oryon_amd/csrc/*.o): unbundles the device code object of each
object file and reads the AMDGPU metadata note.  Prints the kernels that spill; `--all` prints every kernel.  Used by
tests/test_cabi_and_host.py to keep the hot kernels spill-free (a silent spill cost the MX-fp6 screen 12 % in round 3).

    python tools/check_kernel_resources.py [--all]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "host.o")],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fat}", f"--output={co}", "--unbundle"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return []
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        g = lambda key: (re.search(rf"\.{key}:\s*(\S+)", blk) or [None, "0"])[1]
        out.append({"name": g("name"), "vgpr": int(g("vgpr_count")), "agpr": int(blk.split()[0]), "spill": int(g("vgpr_spill_count")),
                    "scratch": int(g("private_segment_fixed_size")), "lds": int(g("group_segment_fixed_size"))})
    return out


def report(show_all=False):
    rows = []
    for obj in sorted(glob.glob(os.path.join(ROOT, "oryon_amd", "csrc", "*.o"))):
        for k in kernels_of(obj):
            k["file"] = os.path.basename(obj)
            rows.append(k)
    for k in rows:
        if show_all or k["spill"] or k["scratch"]:
            print(f"{k['file']:22s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} spill {k['spill']:3d} scratch {k['scratch']:4d} B  {k['name']}")
    return rows


if __name__ == "__main__":
    rows = report("--all" in sys.argv)
    print(f"{len(rows)} kernels, {sum(1 for k in rows if k['spill'])} with spilled registers")
